"""Where does one 1080p P picture go?  enqueue (Python) / GPU / D2H / rANS."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import pad_frame
from bench import synth_sequence
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
dev = torch.device("cuda:0")
i_net, p_net = IntraNoAR(precision=prec).to(dev).eval(), DMC(precision=prec).to(dev).eval()
i_net.update(); p_net.update()
seq = [pad_frame(f) for f in synth_sequence(dev, 6, 1080, 1920, 0)]
dpb = {"ref_frame": i_net.compress(seq[0], 1.0)["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
dpb = p_net.compress(seq[1], dpb, 1.0, 1.0)["dpb"]
torch.cuda.synchronize()
for t in (2, 3, 4):
    t0 = time.perf_counter()
    o = p_net._run(seq[t], dpb, 1.0, 1.0, "compress")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    planes = [o["sym_mv_z"], *o["r_mv"]["sym"], *o["r_mv"]["idx"], o["sym_z"], *o["r_y"]["sym"], *o["r_y"]["idx"]]
    host = [p.cpu().numpy() for p in planes]
    t3 = time.perf_counter()
    ec = p_net.entropy_coder; ec.reset_encoder()
    zs = o["mv_z_hat"]; p_net._encode_factorized("bit_estimator_z_mv", o["sym_mv_z"], 1, 64, zs.H, zs.W)
    for k in (0, 1): p_net._encode_scale(o["r_mv"]["sym"][k], o["r_mv"]["idx"][k])
    zs = o["z_hat"]; p_net._encode_factorized("bit_estimator_z", o["sym_z"], 1, 64, zs.H, zs.W)
    for k in (0, 1): p_net._encode_scale(o["r_y"]["sym"][k], o["r_y"]["idx"][k])
    t4 = time.perf_counter()
    bs = ec.flush_encoder()
    t5 = time.perf_counter()
    dpb = p_net._dpb_out(o)
    print(f"P{t}: enqueue {1e3*(t1-t0):.1f} ms | gpu drain {1e3*(t2-t1):.1f} | D2H(sync copies) {1e3*(t3-t2):.1f} | encode_with_indexes(+D2H again) {1e3*(t4-t3):.1f} | flush {1e3*(t5-t4):.1f} | bytes {len(bs)} | launches {p_net.engine().calls}")
