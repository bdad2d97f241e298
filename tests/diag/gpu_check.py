"""Quick GPU bring-up: HIP path vs CPU oracle on one small I + P + P sequence."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dcvc_ref as R  # noqa: E402
from vcm_ts_amd.dmc import DMC  # noqa: E402
from vcm_ts_amd.intra import IntraNoAR  # noqa: E402
from vcm_ts_amd.params import dmc_spec, intra_spec, seeded_state_dict  # noqa: E402
from vcm_ts_amd.synthetic import frames  # noqa: E402


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    w = int(sys.argv[2]) if len(sys.argv) > 2 else h
    dev = torch.device("cuda:0")
    wd, wi = seeded_state_dict(dmc_spec()), seeded_state_dict(intra_spec())
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    d, i = DMC(precision=prec).to(dev).eval(), IntraNoAR(precision=prec).to(dev).eval()
    fr = frames(0, 3, h, w)
    xs = [torch.from_numpy(fr[t : t + 1]) for t in range(3)]
    with torch.no_grad():
        ro = R.intra_forward(wi, xs[0], 1.0)
        rg = i(xs[0].to(dev), 1.0)
        torch.cuda.synchronize()
        print("intra: x_hat rel", rel(rg["x_hat"], ro["x_hat"]), "bpp", rg["bpp"].item(), ro["bpp"].item(), "mse",
              rg["mse"].item(), ro["mse"].item())
        vg = rg["_views"]
        print("  y_hat rel", rel(vg["y_hat"].nchw(), ro["_inter"]["y_hat"]))
        dpb_o = {"ref_frame": ro["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        dpb_g = {"ref_frame": rg["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        for t in (1, 2):
            po = R.dmc_forward_one_frame(wd, xs[t], dpb_o, 1.0, 1.0)
            t0 = time.time()
            pg = d.forward_one_frame(xs[t].to(dev), dpb_g, 1.0, 1.0)
            torch.cuda.synchronize()
            print(f"P{t}: {time.time() - t0:.3f}s  launches {d.engine().calls}")
            o, v = po["_inter"], pg["_views"]
            for name, gv, ov in (("est_mv", v["est_mv"], o["est_mv"]), ("mv_y_hat", v["mv_y_hat"], o["mv_y_hat"]),
                                 ("mv_hat", v["mv_hat"], o["mv_hat"]), ("c1", v["c1"], o["c1"]), ("c2", v["c2"], o["c2"]),
                                 ("c3", v["c3"], o["c3"]), ("y_hat", v["y_hat"], o["y_hat"]),
                                 ("feature", v["feature"], o["feature"]), ("recon", v["recon"], o["recon"])):
                print(f"   {name:9s} rel {rel(gv.nchw(), ov):.3e}")
            for k in ("bpp_mv_y", "bpp_mv_z", "bpp_y", "bpp_z", "bpp", "me_mse", "mse"):
                print(f"   {k:9s} gpu {pg[k].item():.7f} cpu {po[k].item():.7f}")
            dpb_o, dpb_g = po["dpb"], pg["dpb"]


if __name__ == "__main__":
    main()
