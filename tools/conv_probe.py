"""Time one 3x3 / 7x7 / 1x1 convolution signature at 1088x1920 (or H W): the fp32-activation kernel
(dcvc_conv2d) and, where the layer qualifies, the pre-split kernel (dcvc_conv2d_s16), interleaved in
one process (rounds of 10 launches each, median per variant).
usage: conv_probe.py cin cout ks precision [H W]      cin may be "32,64" for several segments"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.engine import Engine, View
segs = tuple(int(c) for c in sys.argv[1].split(","))
cout, ks, prec = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
H, W = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1088, 1920)
cin = sum(segs)
e = Engine("cuda:0", precision=prec)
xs = []
for i, c in enumerate(segs):
    x = e.buf(f"x{i}", 1, H, W, c, cs=(c + 15) // 16 * 16); x.base.normal_(); xs.append(x)
r = e.buf("r", 1, H, W, cout, cs=(cout + 15) // 16 * 16); r.base.normal_()
o = e.buf("o", 1, H, W, cout, cs=(cout + 15) // 16 * 16)
w = torch.nn.Parameter((torch.randn(cout, cin, ks, ks) * 0.05).cuda()); b = torch.nn.Parameter(torch.zeros(cout).cuda())
pk = e.pack(("p",), w, b, segs, False)
variants = {"f32-act": lambda: e._conv_f32(pk, xs, o, 1, None, 0.01, r, None, None)}
if e.s16_capable(pk):
    s16 = [e.s16_pack(x) for x in xs]
    only = [View(v.base, v.C, 0, geom=(v.N, v.H, v.W, v.cs, v.ptr), fmt="s16") for v in s16]
    o16 = e.buf("o16", 1, H, W, cout, fmt="s16")
    o2 = e.buf("o2", 1, H, W, cout, cs=o.cs, twin=0.01)
    variants["s16->f32"] = lambda: e.conv(pk, only, o, out_slope=0.01, res=r)
    variants["s16->s16"] = lambda: e.conv(pk, only, o16, out_slope=0.01)
    variants["s16->f32+twin"] = lambda: e.conv(pk, only, o2, out_slope=0.01, res=r)
times = {k: [] for k in variants}
for k, fn in variants.items():
    for _ in range(3): fn()
torch.cuda.synchronize()
for rnd in range(7):
    for k, fn in variants.items():
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10): fn()
        ev1.record(); torch.cuda.synchronize()
        times[k].append(ev0.elapsed_time(ev1) / 10)
fl = 2.0 * H * W * cin * cout * ks * ks
for k, t in times.items():
    t = sorted(t); med = t[len(t) // 2]
    print(f"{segs}->{cout} k{ks} {prec} {H}x{W} {k:14s}: median {med:.3f} ms (min {t[0]:.3f})  {fl/med/1e9:.1f} TFLOP/s  frac of 833: {fl/med/1e9/833.3:.3f}")
