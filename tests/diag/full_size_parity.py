"""Direct full-size check (not part of the test-suite: ~1 min of host CPU): one I and one P
picture at 1088x1920 through the CPU oracle and through the HIP path in both precision modes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dcvc_ref as R
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.params import dmc_spec, intra_spec, seeded_state_dict
from vcm_ts_amd.pipeline import pad_frame
from vcm_ts_amd.synthetic import frames
torch.set_num_threads(16)
dev = torch.device("cuda:0")
fr = frames(0, 2, 1080, 1920)
x0, x1 = (pad_frame(torch.from_numpy(fr[t:t+1])) for t in (0, 1))
wd, wi = seeded_state_dict(dmc_spec()), seeded_state_dict(intra_spec())
t0 = time.time()
with torch.no_grad():
    ro = R.intra_forward(wi, x0, 1.0)
    po = R.dmc_forward_one_frame(wd, x1, {"ref_frame": ro["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}, 1.0, 1.0)
print(f"oracle: {time.time()-t0:.1f} s; I bpp {ro['bpp'].item():.6f} mse {ro['mse'].item():.6f}; P bpp {po['bpp'].item():.6f} mse {po['mse'].item():.6f}")
for prec in ("fp32", "fp16x3"):
    i, d = IntraNoAR(precision=prec).to(dev).eval(), DMC(precision=prec).to(dev).eval()
    with torch.no_grad():
        rg = i(x0.to(dev), 1.0)
        pg = d.forward_one_frame(x1.to(dev), {"ref_frame": rg["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}, 1.0, 1.0)
    rel = lambda a, b: abs(a.item() - b.item()) / abs(b.item())
    yq_g = pg["_views"]["r_y"]["y_q"].reshape(1, 68, 120, 96).permute(0, 3, 1, 2).cpu()
    flips = (yq_g != po["_inter"]["y"]["y_q"]).float().mean().item()
    mq_g = pg["_views"]["r_mv"]["y_q"].reshape(1, 68, 120, 64).permute(0, 3, 1, 2).cpu()
    mflips = (mq_g != po["_inter"]["mv"]["y_q"]).float().mean().item()
    print(f"{prec:7s}: I bpp rel {rel(rg['bpp'], ro['bpp']):.2e} mse rel {rel(rg['mse'], ro['mse']):.2e} | "
          f"P bpp rel {rel(pg['bpp'], po['bpp']):.2e} mse rel {rel(pg['mse'], po['mse']):.2e} dPSNR {abs(10*np.log10(pg['mse'].item()/po['mse'].item())):.2e} dB | "
          f"symbols differing: y {flips:.2e}, mv_y {mflips:.2e}")
    del i, d; torch.cuda.empty_cache()
