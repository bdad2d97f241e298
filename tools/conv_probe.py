"""Time one convolution signature at 1088x1920 (or H W): dcvc_conv2d (32x32x16 MFMA, 16-channel chunks) and, where the
layer qualifies, dcvc_conv2d_k32 (16x16x32 MFMA, 32-channel chunks), with and without a residual, interleaved in one
process (rounds of 10 launches each, median per variant).
usage: conv_probe.py cin cout ks precision [H W] [--stride=2]      cin may be "32,64" for several segments"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.engine import Engine
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
stride = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--stride=")), 1)
segs = tuple(int(c) for c in argv[0].split(","))
cout, ks, prec = int(argv[1]), int(argv[2]), argv[3]
H, W = (int(argv[4]), int(argv[5])) if len(argv) > 5 else (1088, 1920)
cin = sum(segs)
e = Engine("cuda:0", precision=prec)
r16c = lambda c: (c + 15) // 16 * 16
xs = []
for i, c in enumerate(segs):
    x = e.buf(f"x{i}", 1, H, W, c, cs=r16c(c)); x.base.normal_(); xs.append(x)
Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
r = e.buf("r", 1, Ho, Wo, cout, cs=r16c(cout)); r.base.normal_()
o = e.buf("o", 1, Ho, Wo, cout, cs=r16c(cout))
w = torch.nn.Parameter((torch.randn(cout, cin, ks, ks) * 0.05).cuda()); b = torch.nn.Parameter(torch.zeros(cout).cuda())
pk = e.pack(("p",), w, b, segs, False)
def run(k32, res):
    e.use_k32 = k32
    e._conv_f32(pk, xs, o, stride, None, 0.01, r if res else None, None, None)
variants = {"conv_mfma": lambda: run(False, False), "conv_mfma +res": lambda: run(False, True)}
e.use_k32 = True
if e.k32_capable(pk, stride, o, r, None, None):
    variants["conv_k32"] = lambda: run(True, False)
    variants["conv_k32 +res"] = lambda: run(True, True)
times = {k: [] for k in variants}
for k, fn in variants.items():
    for _ in range(3): fn()
torch.cuda.synchronize()
for rnd in range(5):
    for k, fn in variants.items():
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10): fn()
        ev1.record(); torch.cuda.synchronize()
        times[k].append(ev0.elapsed_time(ev1) / 10)
fl = 2.0 * Ho * Wo * cin * cout * ks * ks
print(f"{segs}->{cout} k{ks} s{stride} {prec} {H}x{W}")
for k, t in times.items():
    t = sorted(t); med = t[len(t) // 2]
    print(f"  {k:28s}: median {med:.3f} ms (min {t[0]:.3f})  {fl/med/1e9:6.1f} TFLOP/s  frac of 833: {fl/med/1e9/833.3:.3f}")
