"""Does a phase offset between the two GOP streams of the headline run change its rate?  Both streams start their pictures
together (one host thread feeds them round-robin), so their heavy full-resolution layers tend to coincide; a one-off spin on
stream 1 shifts it by a fraction of a picture for the whole GOP.
usage: stagger_probe.py [ms ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_sequence
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import ConcurrentGopEncoder, pad_frame

dev = torch.device("cuda:0")
cenc = ConcurrentGopEncoder(lambda: (IntraNoAR(precision="fp16x3").to(dev).eval(), DMC(precision="fp16x3").to(dev).eval()), gop_size=32, streams=2)
seqs = [[pad_frame(f) for f in synth_sequence(dev, 32, 1080, 1920, seed=k)] for k in range(2)]
cenc.encode_gops([s[:3] for s in seqs], 1.0, 1.0, 1.0)
# calibrate torch.cuda._sleep's unit
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
per_ms = 10_000_000 / e0.elapsed_time(e1)
print(f"_sleep: {per_ms:.0f} cycles per ms")
orig = cenc.encode_gops


def run(stagger_ms):
    if stagger_ms > 0:
        cur = torch.cuda.current_stream(dev)
        cenc.streams[1].wait_stream(cur)
        with torch.cuda.stream(cenc.streams[1]):
            torch.cuda._sleep(int(stagger_ms * per_ms))
    torch.cuda.synchronize(dev) if stagger_ms == 0 else None
    t0 = time.time()
    res = cenc.encode_gops(seqs, 1.0, 1.0, 1.0)
    torch.cuda.synchronize(dev)
    dt = time.time() - t0
    return 64 / dt, [len(b"".join(c[2] for c in r[0])) for r in res]


vals = [float(v) for v in sys.argv[1:]] or [0, 6, 12, 18, 0, 12]
run(0)
for ms in vals:
    fps = [run(ms)[0] for _ in range(3)]
    print(f"stagger {ms:5.1f} ms: {sorted(fps)[1]:.2f} frames/s (median of 3; the spin itself is inside the time)")
