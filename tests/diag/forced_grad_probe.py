"""Developer diagnostic (GPU box): which loss term carries the deviation of a parameter's gradient?

BASELINE configs[2]'s shape (batch 4 of 256x256) with the gradient fixture's inputs: the oracle (torch-CPU autograd,
pinned to the reference's gradients at 6e-7, tests/test_oracle_train.py) runs free, the HIP path is FORCED to the
oracle's rounded integers (DMC._forced), so both sides see the same symbols; then every loss term alone
(bpp_y, bpp_z, bpp_mv_y, bpp_mv_z, mse, me_mse) is back-propagated on both sides and the named tensors are compared.

    python tests/diag/forced_grad_probe.py [fp32|fp16x3] [tensor substring ...]
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
from vcm_ts_amd.params import dmc_spec, seeded_state_dict  # noqa: E402
from vcm_ts_amd.synthetic import frames  # noqa: E402

TERMS = ("bpp_y", "bpp_z", "bpp_mv_y", "bpp_mv_z", "mse", "me_mse")


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    watch = sys.argv[2:] or ["contextual_hyper_prior_encoder.0.weight", "contextual_hyper_prior_encoder.4.weight",
                             "mv_hyper_prior_encoder.0.weight", "y_prior_fusion.0.weight", "contextual_encoder.conv1.weight"]
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
    forced = {"mv_z": torch.round(o["mv_z_hat"].detach()), "z": torch.round(o["z_hat"].detach()),
              "mv": o["mv"]["y_q"].detach(), "y": o["y"]["y_q"].detach()}
    for k in ("mv_z", "z", "mv", "y"):
        same = np.array_equal(forced[k].numpy(), fx["s0_rounded_" + k].astype(np.float32))
        print(f"oracle's rounded {k} equals the reference's: {same}")
    m = DMC(precision=precision).to(dev).train()
    for p in m.parameters():
        p.requires_grad_(True)
    m._noise_override = noise
    m._forced = forced
    params = dict(m.named_parameters())
    dpb_g = {"ref_frame": x0.to(dev), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    print(f"{'term':10s} " + " ".join(f"{n[-38:]:>40s}" for n in watch))
    for term in TERMS + ("all",):
        for v in w.values():
            v.grad = None
        lo = torch.mean(ro[term]) if term != "all" else torch.mean(ro["bpp"] + 85.0 * ro["mse"] + 10.0 * ro["me_mse"])
        lo.backward(retain_graph=True)
        m.zero_grad(set_to_none=True)
        rg = m.forward_one_frame(x1.to(dev), dpb_g, q_mv.to(dev), q_y.to(dev))
        lg = torch.mean(rg[term]) if term != "all" else torch.mean(rg["bpp"] + 85.0 * rg["mse"] + 10.0 * rg["me_mse"])
        lg.backward()
        cells = []
        for n in watch:
            go, gg = w[n].grad, params[n].grad
            if go is None or gg is None:
                cells.append(f"{'-':>40s}")
                continue
            gg = gg.cpu().double()
            go = go.double()
            cells.append(f"|ref| {float(go.norm()):9.3e} rel {float((gg - go).norm() / (go.norm() + 1e-30)):8.2e}".rjust(40))
        print(f"{term:10s} " + " ".join(cells) + f"   loss rel {abs(lg.item() - lo.item()) / abs(lo.item()):.1e}")
        if term == os.environ.get("PROBE_LIST_TERM", "bpp_y"):  # every tensor this term reaches, in module order
            for n, prm in params.items():
                go, gg = w[n].grad, prm.grad
                if go is None or gg is None or float(go.norm()) == 0.0:
                    continue
                gg, go = gg.cpu().double(), go.double()
                print(f"      {n:60s} |ref| {float(go.norm()):9.3e}  rel {float((gg - go).norm() / go.norm()):8.2e}  "
                      f"norm ratio {float(gg.norm() / go.norm()):.4f}")


if __name__ == "__main__":
    main()
