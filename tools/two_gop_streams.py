"""Two GOPs of one sequence encoded concurrently on one GPU (two HIP streams, two codec instances,
one host thread interleaving the pictures): do the small kernels of one GOP fill the gaps of the other?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, pad_frame
from vcm_ts_amd import stream as S
from bench import synth_sequence
dev = torch.device("cuda:0")
gop = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
encs = [GopEncoder(IntraNoAR(precision="fp16x3").to(dev).eval(), DMC(precision="fp16x3").to(dev).eval(), gop) for _ in range(K)]
seqs = [[pad_frame(f) for f in synth_sequence(dev, gop, 1080, 1920, seed=k)] for k in range(K)]
streams = [torch.cuda.Stream(dev) for _ in range(K)]

def one_at_a_time():
    return [encs[0].encode_gop(seqs[k], 1.0, 1.0, 1.0)[0] for k in range(K)]

def concurrent():
    out = [[] for _ in range(K)]
    dpb = [None] * K
    prev = [None] * K
    for t in range(gop):
        for k in range(K):
            e = encs[k]
            with torch.cuda.stream(streams[k]):
                if t == 0:
                    r = e.i_net.compress(seqs[k][t], 1.0, defer=True)
                    dpb[k] = {"ref_frame": r["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
                else:
                    r = e.p_net.compress(seqs[k][t], dpb[k], 1.0, 1.0, defer=True)
                    dpb[k] = r["dpb"]
            if prev[k] is not None:
                out[k].append(prev[k].finish())
            prev[k] = r["pending"]
    for k in range(K):
        out[k].append(prev[k].finish())
    return out

for fn in (one_at_a_time, concurrent):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); res = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{fn.__name__:16s}: {K * gop / dt:6.2f} frames/s  ({K} GOPs of {gop})", flush=True)
    if fn is one_at_a_time:
        ref = [[c[2] for c in g] for g in res]
    else:
        print("   payloads identical to sequential encode:", all(a == b for g1, g2 in zip(ref, res) for a, b in zip(g1, g2)))
