"""Which parameter gradients differ between identical training-mode runs of two pictures (developer diagnostic;
the assertion lives in tests/test_gpu_backward.py::test_training_step_is_bit_reproducible).
usage: repro_check.py [N size]   (round 2 used it to bisect the run-to-run difference of warp_bwd_kernel, backward.hip)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.synthetic import frames

dev = torch.device("cuda:0")
m = DMC(precision="fp16x3").to(dev).train()
for p in m.parameters():
    p.requires_grad_(True)
N, size = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 128)
fr = frames(9, N * 3, size, size)
x0, x1, x2 = (torch.from_numpy(fr[k * N:(k + 1) * N]).to(dev) for k in range(3))
g = torch.Generator().manual_seed(3)
m._noise_override = {"y": torch.rand(N, 96, size // 16, size // 16, generator=g) - 0.5,
                     "mv_y": torch.rand(N, 64, size // 16, size // 16, generator=g) - 0.5,
                     "z": torch.rand(N, 64, size // 64, size // 64, generator=g) - 0.5,
                     "mv_z": torch.rand(N, 64, size // 64, size // 64, generator=g) - 0.5}


def run():
    grads = []
    dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for x in (x1, x2):
        m.zero_grad(set_to_none=True)
        out = m.forward_one_frame(x, dpb, 1.0, 1.0)
        loss = torch.mean(out["bpp"] + 256.0 * out["mse"] + out["me_mse"])
        loss.backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        dpb = {k: v.detach() for k, v in out["dpb"].items()}
    return grads


runs = [run() for _ in range(3)]
for j in (1, 2):
    for i, (ga, gb) in enumerate(zip(runs[0], runs[j])):
        bad = [(k, float((ga[k] - gb[k]).abs().max()), float(ga[k].abs().max()), tuple(ga[k].shape)) for k in ga if not torch.equal(ga[k], gb[k])]
        print(f"run 0 vs {j}, picture {i}: {len(bad)} of {len(ga)} differ")
        for b in bad[:12]:
            print("   ", b)

