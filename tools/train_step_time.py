"""Time the training step of BASELINE config 3 (trainer.py, batch 4 of 256x256 clips, bpp + MSE
loss, `single` mode: one optimiser step per P picture = forward_one_frame + backward + AdamW)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dcvc_hem import build_model, make_cfg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
dev = torch.device("cuda:0")
lam = (85.0, 170.0, 380.0, 840.0)[: a.batch] if a.batch <= 4 else tuple(85.0 * (i + 1) for i in range(a.batch))
model = build_model(make_cfg(lambdas=lam), precision=a.precision).to(dev).train()
model.activate_modules_all()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(0)
T = a.warmup + a.steps + 1
clip = torch.rand(a.batch, T, 3, a.size, a.size, generator=g).to(dev)


def run(t0, n):
    dpb = {"ref_frame": clip[:, t0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for t in range(t0 + 1, t0 + 1 + n):
        opt.zero_grad()
        r = model("single_multi", clip[:, t], clip[:, t], "mse", ["bpp"], perceptual_loss=False, dpb=dpb)
        r["loss_to_opt"].backward()
        opt.step()
        dpb = r["dpb"]
    return r


run(0, a.warmup)
torch.cuda.synchronize()
t0 = time.time()
r = run(a.warmup, a.steps)
torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
out = {"workload": f"trainer step: batch {a.batch} x {a.size}x{a.size}, bpp+MSE, AdamW, single mode", "precision": a.precision,
       "ms_per_step": round(dt * 1e3, 2), "frames_per_s": round(a.batch / dt, 2), "loss": float(r["loss_to_opt"]),
       "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
print(json.dumps(out))
