"""cProfile of the host side of training steps (batch 4 x 256x256): where the Python time between launches goes.
usage (on the GPU box): python3 tools/py_profile_train.py [steps] > gpurun_out/py_profile_train.txt"""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dcvc_hem import build_model, make_cfg

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
model = build_model(make_cfg(lambdas=(85.0, 170.0, 380.0, 840.0)), precision="fp16x3").to(dev).train()
model.activate_modules_all()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
clip = torch.rand(4, steps + 4, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)


def run(t0, n):
    dpb = {"ref_frame": clip[:, t0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for t in range(t0 + 1, t0 + 1 + n):
        opt.zero_grad()
        r = model("single_multi", clip[:, t], clip[:, t], "mse", ["bpp"], perceptual_loss=False, dpb=dpb)
        r["loss_to_opt"].backward()
        opt.step()
        dpb = r["dpb"]


run(0, 2)
torch.cuda.synchronize()
t0 = time.time()
run(2, steps)
torch.cuda.synchronize()
print(f"{steps} steps: {(time.time() - t0) / steps * 1e3:.2f} ms per step (wall, unprofiled)")
t0 = time.time()
run(2, steps)
t_issue = time.time() - t0
torch.cuda.synchronize()
print(f"host time to ISSUE {steps} steps: {t_issue / steps * 1e3:.2f} ms per step; incl. drain {(time.time() - t0) / steps * 1e3:.2f}")
from vcm_ts_amd import grad as G
bpr = cProfile.Profile()
_orig_backward = G.Tape.backward


def _profiled_backward(self):  # autograd runs this in its own thread: it needs its own profiler
    bpr.enable()
    try:
        return _orig_backward(self)
    finally:
        bpr.disable()


pr = cProfile.Profile()
pr.enable()
run(2, steps)
pr.disable()
torch.cuda.synchronize()
G.Tape.backward = _profiled_backward
run(2, steps)
torch.cuda.synchronize()
G.Tape.backward = _orig_backward
s = io.StringIO()
pstats.Stats(bpr, stream=s).sort_stats("tottime").print_stats(30)
print("==== Tape.backward (autograd thread) ====")
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).print_callers("cpu")
print(s.getvalue())
