"""GPU busy/idle per picture over one pipelined GOP encode (events at the end of every picture)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, pad_frame
from bench import synth_sequence
dev = torch.device("cuda:0")
i_net, p_net = IntraNoAR(precision="fp16x3").to(dev).eval(), DMC(precision="fp16x3").to(dev).eval()
enc = GopEncoder(i_net, p_net, 32)
seq = [pad_frame(f) for f in synth_sequence(dev, 32, 1080, 1920, 0)]
enc.encode_gop(seq, 1.0, 1.0, 1.0)
torch.cuda.synchronize()
evs = []
host = []
orig_p, orig_i = p_net.compress, i_net.compress
def wrap(fn):
    def f(*a, **k):
        t0 = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        r = fn(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        evs.append((e0, e1)); host.append((t0, time.perf_counter()))
        return r
    return f
p_net.compress, i_net.compress = wrap(orig_p), wrap(orig_i)
t0 = time.perf_counter()
enc.encode_gop(seq, 1.0, 1.0, 1.0)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"GOP wall {1e3*(t1-t0):.1f} ms")
base = evs[0][0]
for k, ((e0, e1), (h0, h1)) in enumerate(zip(evs, host)):
    if k < 4 or k > 28:
        print(f"pic {k:2d}: gpu start {base.elapsed_time(e0):8.1f} end {base.elapsed_time(e1):8.1f} dur {e0.elapsed_time(e1):6.1f} | host enqueue at {1e3*(h0-t0):8.1f} took {1e3*(h1-h0):5.1f}")
print("sum of gpu picture durations %.1f ms" % sum(a.elapsed_time(b) for a, b in evs))
