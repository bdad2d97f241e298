"""Per-convolution-launch timing of one 1088x1920 P picture (HIP events), grouped by signature."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import pad_frame
from bench import synth_sequence
dev = torch.device("cuda:0")
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
i_net, p_net = IntraNoAR(precision=prec).to(dev).eval(), DMC(precision=prec).to(dev).eval()
i_net.update(); p_net.update()
seq = [pad_frame(f) for f in synth_sequence(dev, 3, 1080, 1920, 0)]
dpb = {"ref_frame": i_net.compress(seq[0], 1.0)["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
dpb = p_net.compress(seq[1], dpb, 1.0, 1.0)["dpb"]
which = sys.argv[1] if len(sys.argv) > 1 else "P"
net = p_net if which == "P" else i_net
e = net.engine(); e.profile = {}; e.profile_detail = []
if which == "P":
    p_net.compress(seq[2], dpb, 1.0, 1.0)
else:
    i_net.compress(seq[0], 1.0)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for a, b, fl, sig, _note in e.profile_detail:
    d = agg.setdefault(sig, [0, 0.0, 0.0]); d[0] += 1; d[1] += a.elapsed_time(b); d[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"total conv time {tot:.2f} ms, {sum(v[0] for v in agg.values())} launches, {sum(v[2] for v in agg.values())/tot/1e9:.1f} TFLOP/s")
for sig, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{sig:48s} n={n:3d} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9:7.1f} TFLOP/s")
