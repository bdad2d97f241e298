"""Time one 3x3 convolution signature at 1088x1920 (or H W): the fp32-activation kernel (dcvc_conv2d) and, where
the layer qualifies, the pre-split kernel (dcvc_conv2d_s16) in its output / residual formats, interleaved in one
process (rounds of 10 launches each, median per variant).  --ablate adds the probe builds of
tools/probes/conv_s16_probe.hip (no DMA / no MFMA / no epilogue) for the all-s16 variant.
usage: conv_probe.py cin cout ks precision [H W] [--ablate] [--stride=2]      cin may be "32,64" for several segments"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd import lib
from vcm_ts_amd.engine import Engine, View
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
ablate = "--ablate" in sys.argv
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
variants = {"f32-act+res": lambda: e._conv_f32(pk, xs, o, stride, None, 0.01, r, None, None),
            "f32-act": lambda: e._conv_f32(pk, xs, o, stride, None, 0.01, None, None, None)}
if e.s16_capable(pk, stride):
    only = []
    for x in xs:
        v = e.s16_pack(x)
        only.append(View(v.base, v.C, 0, geom=(v.N, v.H, v.W, v.cs, v.ptr), fmt="s16"))
    rv = e.s16_pack(r)
    r16 = View(rv.base, rv.C, 0, geom=(rv.N, rv.H, rv.W, rv.cs, rv.ptr), fmt="s16")
    o16 = e.buf("o16", 1, H, W, cout, fmt="s16")
    o2 = e.buf("o2", 1, H, W, cout, cs=o.cs, twin=0.01)
    variants["s16->s16"] = lambda: e.conv(pk, only, o16, out_slope=0.01)
    variants["s16->s16+res16"] = lambda: e.conv(pk, only, o16, out_slope=0.01, res=r16)
    variants["s16->f32+res32"] = lambda: e.conv(pk, only, o, out_slope=0.01, res=r)
    variants["s16->f32,s16+res16"] = lambda: e.conv(pk, only, o2, out_slope=0.01, res=r16)
    if ablate and pk.Cout_pad % 64 == 0:
        P = C.CDLL(os.path.join(ROOT, "tools", "probes", "libconv_s16_probe.so"))
        P.dcvc_conv2d_s16_probe.argtypes = [C.POINTER(lib.ConvS16Args), C.c_int, C.c_void_p]
        q = e.pack_s16(pk)
        def mk(res):
            a = lib.ConvS16Args()
            for i, s in enumerate(only):
                a.seg[i].ptr, a.seg[i].C, a.seg[i].cs = s.ptr, s.C, s.cs
            a.nseg, a.N, a.H, a.W = len(only), 1, H, W
            a.wpack, a.bpack, a.ks, a.Cout, a.Cout_pad = q.w.data_ptr(), q.b.data_ptr(), 3, q.Cout, q.Cout_pad
            a.out_act, a.out_slope = 1, 0.01
            a.out16, a.out16_cs = o16.ptr, o16.cs
            if res:
                a.res, a.res_cs, a.res_fmt = r16.ptr, r16.cs, 1
            return a
        for res in (False, True):
            a = mk(res)
            for base, tag in ((0, "probe"),):
                for probe, nm in ((0, "full"), (1, "noDMA"), (2, "noMFMA"), (4, "noEPI"), (3, "noDMA,noMFMA"), (5, "noDMA,noEPI"), (6, "noMFMA,noEPI")):
                    variants[f"{tag}[{nm}]{'+res16' if res else ''}"] = (lambda a=a, probe=base + probe: lib.check(P.dcvc_conv2d_s16_probe(C.byref(a), probe, e.stream()), "probe"))
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
