"""Is a convolution kernel bound by its structure or by board power?  The same launch on random operands and on
all-zero operands (no toggling: the chip holds its full clock), interleaved in one process.  If the zero-data time is
much lower, the kernel is clock-limited by power on real data (MI355X_MICROARCH.md "DVFS give-back" item 1) and the
zero-data time is what its instruction stream costs at 2.4 GHz; if the two agree, stalls bound it.
usage: conv_data_probe.py [cin cout ks [H W]]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.engine import Engine
argv = sys.argv[1:]
cin, cout, ks = (int(argv[0]), int(argv[1]), int(argv[2])) if len(argv) >= 3 else (64, 64, 3)
H, W = (int(argv[3]), int(argv[4])) if len(argv) >= 5 else (1088, 1920)
e = Engine("cuda:0", precision="fp16x3")
bufs = {}
for tag in ("rand", "zero"):
    x = e.buf(f"x{tag}", 1, H, W, cin); r = e.buf(f"r{tag}", 1, H, W, cout); o = e.buf(f"o{tag}", 1, H, W, cout)
    if tag == "rand":
        x.base.normal_(); r.base.normal_()
        w = torch.randn(cout, cin, ks, ks) * 0.05
    else:
        x.base.zero_(); r.base.zero_()
        w = torch.zeros(cout, cin, ks, ks)
    pk = e.pack((tag,), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(torch.zeros(cout).cuda()), (cin,), False)
    bufs[tag] = (x, r, o, pk)
variants = {}
for k32 in (True, False):
    for tag in ("rand", "zero"):
        for res in (False, True):
            def fn(k32=k32, tag=tag, res=res):
                x, r, o, pk = bufs[tag]
                e.use_k32 = k32
                e._conv_f32(pk, [x], o, 1, None, 0.01, r if res else None, None, None)
            variants[f"{'k32 16x16x32' if k32 else 'conv_mfma 32x32x16'} {tag}{' +res' if res else ''}"] = fn
for fn in variants.values():
    for _ in range(3): fn()
torch.cuda.synchronize()
t_end = torch.cuda.Event(enable_timing=True)
times = {k: [] for k in variants}
for _ in range(40):  # settle the clock under load
    for fn in variants.values(): fn()
torch.cuda.synchronize()
for rnd in range(7):
    for k, fn in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 20)
fl = 2.0 * H * W * cin * cout * ks * ks
print(f"({cin},)->{cout} k{ks} s1 fp16x3 {H}x{W}; 7 rounds x 20 launches, interleaved")
for k, t in times.items():
    t = sorted(t); med = t[len(t) // 2]
    print(f"  {k:38s}: median {med:.3f} ms (min {t[0]:.3f})  {fl/med/1e9:6.1f} TFLOP/s  frac of 833: {fl/med/1e9/833.3:.3f}")
