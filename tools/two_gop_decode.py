"""Two GOPs decoded concurrently on one GPU: two host threads, two HIP streams, two codec instances."""
import os, sys, time, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, pad_frame
from bench import synth_sequence
dev = torch.device("cuda:0")
gop = int(sys.argv[1]) if len(sys.argv) > 1 else 12
K = 2
for coder in ("host", "device"):
    encs = [GopEncoder(IntraNoAR(precision="fp16x3").to(dev).eval(), DMC(precision="fp16x3").to(dev).eval(), gop, coder=coder) for _ in range(K)]
    seqs = [[pad_frame(f) for f in synth_sequence(dev, gop, 1080, 1920, seed=k)] for k in range(K)]
    coded = [encs[k].encode_gop(seqs[k], 1.0, 1.0, 1.0)[0] for k in range(K)]
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    torch.cuda.synchronize()
    def seq_run():
        return [encs[0].decode_gop(coded[k], 1080, 1920)[-1].clone() for k in range(K)]
    def thr_run():
        out = [None] * K
        def work(k):
            with torch.cuda.stream(streams[k]):
                out[k] = encs[k].decode_gop(coded[k], 1080, 1920)[-1].clone()
        ts = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        [t.start() for t in ts]; [t.join() for t in ts]
        return out
    for fn in (seq_run, thr_run):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if fn is seq_run: ref = r
        print(f"{coder:6s} {fn.__name__:8s}: {K*gop/dt:6.2f} frames/s" + ("" if fn is seq_run else f"  identical: {all(torch.equal(a,b) for a,b in zip(ref,r))}"), flush=True)
    del encs
    torch.cuda.empty_cache()
