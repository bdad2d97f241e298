"""Timing probe: a few 1080p pictures through compress, per-phase wall time."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, pad_frame
from bench import synth_sequence
dev = torch.device("cuda:0")
i_net, p_net = IntraNoAR().to(dev).eval(), DMC().to(dev).eval()
enc = GopEncoder(i_net, p_net, 32)
seq = [pad_frame(f) for f in synth_sequence(dev, 6, 1080, 1920, 0)]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    r = i_net.compress(seq[0], 1.0); torch.cuda.synchronize(); t1 = time.time()
    print(f"I: {t1-t0:.3f}s bytes {len(r['bit_stream'])}")
    dpb = {"ref_frame": r["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for t in range(1, 6):
        t0 = time.time(); r = p_net.compress(seq[t], dpb, 1.0, 1.0); torch.cuda.synchronize(); t1 = time.time()
        dpb = r["dpb"]
        print(f"P{t}: {t1-t0:.3f}s bytes {len(r['bit_stream'])}")
print("HBM reserved by engine buffers: P %.2f GB, I %.2f GB" % (p_net.engine().bytes_reserved()/1e9, i_net.engine().bytes_reserved()/1e9))
# estimate-path only (no entropy coding) for kernel time
t0=time.time()
for t in range(1,6):
    r = p_net.forward_one_frame(seq[t], dpb, 1.0, 1.0); dpb = r["dpb"]
torch.cuda.synchronize(); print("forward_one_frame avg %.3f s" % ((time.time()-t0)/5))
