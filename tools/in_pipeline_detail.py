"""Every convolution launch of one 1088x1920 P picture in issue order: layer, flags, HIP-event time, TFLOP/s.
Then the full-resolution 64->64 3x3 launches re-timed ALONE (same buffers, same flags, 10 back-to-back launches each)
to separate what a launch costs by itself from what it costs where it sits in the picture.
usage: in_pipeline_detail.py [precision] [filter substring]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import pad_frame
from bench import synth_sequence
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
flt = sys.argv[2] if len(sys.argv) > 2 else "(64,)->64 1088x1920"
i_net, p_net = IntraNoAR(precision=prec).to(dev).eval(), DMC(precision=prec).to(dev).eval()
i_net.update(); p_net.update()
p_net.fork_features = False  # every launch on one stream: a launch's events then time that launch alone
seq = [pad_frame(f) for f in synth_sequence(dev, 4, 1080, 1920, 0)]
dpb = {"ref_frame": i_net.compress(seq[0], 1.0)["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
dpb = p_net.compress(seq[1], dpb, 1.0, 1.0)["dpb"]
dpb = p_net.compress(seq[2], dpb, 1.0, 1.0)["dpb"]
e = p_net.engine()
# record the launches (closures) so that they can be replayed alone afterwards
replay = []
orig = e._launch_conv
def rec(launch, pk, s0, Ho, Wo, stride, res, res2, tag="", note=""):
    replay.append(launch)
    return orig(launch, pk, s0, Ho, Wo, stride, res, res2, tag, note)
e._launch_conv = rec
e.profile = {}; e.profile_detail = []
p_net.compress(seq[3], dpb, 1.0, 1.0)
torch.cuda.synchronize()
e._launch_conv = orig
det = e.profile_detail
e.profile = None; e.profile_detail = None
tot = sum(a.elapsed_time(b) for a, b, *_ in det)
print(f"# {len(det)} convolution launches, {tot:.2f} ms in all ({prec})")
rows = []
for i, ((a, b, fl, sig, note), fn) in enumerate(zip(det, replay)):
    ms = a.elapsed_time(b)
    rows.append((i, sig, note, ms, fl, fn))
    print(f"{i:3d} {ms:7.3f} ms {fl/ms/1e9:6.1f} TF  {sig:40s} {note}")
print(f"\n# launches matching '{flt}' replayed alone: 3 warm-up + 5 rounds of 10 back-to-back launches, median")
for i, sig, note, ms, fl, fn in rows:
    if flt not in sig:
        continue
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    print(f"{i:3d} in picture {ms:7.3f} ms | alone {ts[2]:7.3f} ms (min {ts[0]:.3f}) {fl/ts[2]/1e9:6.1f} TF  {note}")
