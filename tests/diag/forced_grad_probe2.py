"""Developer diagnostic (GPU box), second stage of forced_grad_probe.py: for ONE loss term, the gradient with respect to
the INTERMEDIATE tensors of the picture (oracle: retain_grad on its intermediates; HIP path: the tape's gradient
buffers), from the loss backwards -- where along the chain does the deviation enter?

    python tests/diag/forced_grad_probe2.py [fp32|fp16x3] [term]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import dcvc_ref as R  # noqa: E402
from tests.util import golden  # noqa: E402
from vcm_ts_amd.dmc import DMC  # noqa: E402
from vcm_ts_amd.grad import Tape  # noqa: E402
from vcm_ts_amd.params import dmc_spec, seeded_state_dict  # noqa: E402
from vcm_ts_amd.synthetic import frames  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30))


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    term = sys.argv[2] if len(sys.argv) > 2 else "bpp_y"
    fx = golden("train_256_b4")
    N, size = int(fx["meta"][0]), int(fx["meta"][1])
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    dev = torch.device("cuda:0")
    fr = frames(3, N * 3, size, size)
    x0, x1 = torch.from_numpy(fr[0:N]), torch.from_numpy(fr[N:2 * N])
    q_mv = torch.from_numpy(fx["q_mv"]).float().view(N, 1, 1, 1)
    q_y = torch.from_numpy(fx["q_y"]).float().view(N, 1, 1, 1)
    noise = {k: torch.from_numpy(fx["s0_noise_" + k]) for k in ("y", "mv_y", "z", "mv_z")}
    w = {k: v.clone().requires_grad_() for k, v in seeded_state_dict(dmc_spec()).items()}
    with R.training_mode():
        ro = R.dmc_forward_one_frame(w, x1, {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}, q_mv, q_y,
                                     noise=noise)
    o = ro["_inter"]
    inter = {"y.sp_out": o["y"]["sp_out"], "y.sp_in": o["y"]["sp_in"], "y_in": o["y_in"], "y_fusion": o["y_fusion"], "y.y_res": o["y"]["y_res"], "y.scales_hat": o["y"]["scales_hat"], "y_hat": o["y_hat"], "z": o["z"], "c3": o["c3"],
             "c2": o["c2"], "c1": o["c1"], "mv_hat": o["mv_hat"], "mv_y_hat": o["mv_y_hat"], "mv.y_res": o["mv"]["y_res"],
             "mv.scales_hat": o["mv"]["scales_hat"], "mv_z": o["mv_z"], "est_mv": o["est_mv"]}
    for t in inter.values():
        t.retain_grad()
    torch.mean(ro[term]).backward()
    forced = {"mv_z": torch.round(o["mv_z_hat"].detach()), "z": torch.round(o["z_hat"].detach()),
              "mv": o["mv"]["y_q"].detach(), "y": o["y"]["y_q"].detach()}
    m = DMC(precision=precision).to(dev).train()
    m._noise_override = noise
    m._forced = forced
    e = m.engine()
    tape = Tape(e)
    tape.dpb_grad = set()
    with torch.no_grad():
        g, sums = m._train_frame(tape, x1.to(dev), {"ref_frame": x0.to(dev), "ref_feature": None, "ref_y": None, "ref_mv_y": None},
                                 q_mv.to(dev), q_y.to(dev))
        pix = size * size
        up_name = {"bpp_y": "bits_y", "bpp_z": "bits_z", "bpp_mv_y": "bits_mv_y", "bpp_mv_z": "bits_mv_z", "mse": "sq", "me_mse": "me_sq"}[term]
        tape.up[up_name] = torch.full((N,), 1.0 / (N * pix), device=dev)
        print(f"{term}: oracle {float(torch.mean(ro[term])):.8f}  hip {float(sums[up_name].sum()) / (N * pix):.8f}")
        tape.backward()
        views = {"y.sp_out": g["r_y"]["spatial"], "y.sp_in": g["r_y"]["params"], "y_in": g["y"], "y_fusion": g["fusion_y"], "y_hat": g["y_hat"], "z": g["z"], "c3": g["c3"], "c2": g["c2"], "c1": g["c1"], "mv_hat": g["mv_hat"],
                 "mv_y_hat": g["mv_y_hat"], "mv_z": g["mv_z"], "est_mv": g["est_mv"]}
        dense = {"y.y_res": g["r_y"]["y_res"], "y.scales_hat": g["r_y"]["scales_hat"], "mv.y_res": g["r_mv"]["y_res"],
                 "mv.scales_hat": g["r_mv"]["scales_hat"]}
        print(f"{'tensor':16s} {'|ref grad|':>12s} {'rel diff':>10s} {'norm ratio':>10s}")
        for name, t in inter.items():
            go = t.grad
            if go is None:
                print(f"{name:16s} no oracle gradient")
                continue
            if name in views:
                gv = tape.grad(views[name], create=False)
                gg = None if gv is None else e.to_nchw(gv)
            else:
                d = tape.dense.get(dense[name].data_ptr())
                gg = None if d is None else d.view(go.shape[0], go.shape[2], go.shape[3], go.shape[1]).permute(0, 3, 1, 2)
            if gg is None:
                print(f"{name:16s} |ref| {float(go.norm()):.3e}  no HIP gradient")
                continue
            r, nr = rel(gg, go)
            print(f"{name:16s} {float(go.norm()):12.4e} {r:10.2e} {nr:10.4f}")
            if name == "y.y_res":  # element-wise: where do the two direct gradients of the rate differ?
                a, b = gg.cpu().double().reshape(-1), go.double().reshape(-1)
                d = (a - b).abs()
                yb = (o["y"]["y_res"] + noise["y"]).detach().reshape(-1)
                sh = o["y"]["scales_hat"].detach().reshape(-1)
                gyb = g["r_y"]["y_res"].view(go.shape[0], go.shape[2], go.shape[3], go.shape[1]).permute(0, 3, 1, 2).cpu().reshape(-1)
                gsh = g["r_y"]["scales_hat"].view(go.shape[0], go.shape[2], go.shape[3], go.shape[1]).permute(0, 3, 1, 2).cpu().reshape(-1)
                print(f"    forward inputs: y_res max abs diff {float((gyb - o['y']['y_res'].detach().reshape(-1)).abs().max()):.3e}, "
                      f"scales_hat max abs diff {float((gsh - sh).abs().max()):.3e}")
                print(f"    elements: {d.numel()}, differing by > 1e-3 of the largest gradient: {int((d > 1e-3 * b.abs().max()).sum())}")
                for i in torch.topk(d, 12).indices.tolist():
                    print(f"      i {i:7d} y_bit {float(yb[i]):+10.4f} scale {float(sh[i]):10.3e} (hip {float(gsh[i]):10.3e})  grad oracle {float(b[i]):+.4e} "
                          f"hip {float(a[i]):+.4e}")
                frac = float((d * d).sum() / ((a - b).norm() ** 2))
                top = torch.topk(d, 100).values
                print(f"    share of the squared difference held by the 100 worst elements: {float((top * top).sum() / (d * d).sum()):.3f}")
            if name in ("y.sp_in", "y_fusion", "y.sp_out"):
                C = go.shape[1]
                step = C // (4 if name != "y_fusion" else 3)
                for k in range(0, C, step):
                    r, nr = rel(gg[:, k:k + step], go[:, k:k + step])
                    print(f"    channels {k:3d}..{k + step - 1:3d} {float(go[:, k:k + step].norm()):12.4e} {r:10.2e} {nr:10.4f}")


if __name__ == "__main__":
    main()
