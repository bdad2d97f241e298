"""Does running a ResBlock's two convolutions BAND BY BAND keep the intermediate in the 256 MB Infinity Cache?
x -> conv1 (LeakyReLU in / out) -> a -> conv2 (+ x) -> out at 1088x1920, 64 channels, fp16x3: as two whole-picture launches
(a = 535 MB goes to HBM and comes back) against B bands, conv1 one band ahead of conv2 (conv2 of band k needs the first
row of a's band k+1).  Interleaved rounds in one process.   usage: band_probe.py [channels]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.engine import Engine
C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 1088, 1920
e = Engine("cuda:0", precision="fp16x3")
x = e.buf("x", 1, H, W, C_); x.base.normal_()
a = e.buf("a", 1, H, W, C_); o = e.buf("o", 1, H, W, C_); o2 = e.buf("o2", 1, H, W, C_)
mk = lambda t: e.pack((t,), torch.nn.Parameter((torch.randn(C_, C_, 3, 3) * 0.04).cuda()), torch.nn.Parameter(torch.zeros(C_).cuda()), (C_,), False)
p1, p2 = mk("c1"), mk("c2")
nty = (H + 7) // 8
def whole(out):
    e.conv(p1, [x], a, in_slope=0.01, out_slope=0.01)
    e.conv(p2, [a], out, res=x)
def banded(out, nb):
    rows = (nty + nb - 1) // nb
    bands = [(r, min(rows, nty - r)) for r in range(0, nty, rows)]
    e.conv(p1, [x], a, in_slope=0.01, out_slope=0.01, band=bands[0])
    for k, b in enumerate(bands):
        if k + 1 < len(bands):
            e.conv(p1, [x], a, in_slope=0.01, out_slope=0.01, band=bands[k + 1])
        e.conv(p2, [a], out, res=x, band=b)
whole(o); torch.cuda.synchronize()
for nb in (4, 8, 17):
    o2.base.zero_(); banded(o2, nb); torch.cuda.synchronize()
    assert torch.equal(o.base, o2.base), f"banded ({nb}) result differs"
variants = {"2 whole-picture launches": lambda: whole(o)}
for nb in (2, 4, 8, 17, 34):
    variants[f"{nb} bands ({2 * nb} launches)"] = (lambda nb=nb: banded(o2, nb))
times = {k: [] for k in variants}
for _ in range(20):
    for fn in variants.values(): fn()
torch.cuda.synchronize()
for rnd in range(7):
    for k, fn in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 10)
print(f"ResBlock pair, {C_} channels, {H}x{W}, fp16x3 (conv_k32); banded results bit-identical to whole-picture launches")
for k, t in times.items():
    t = sorted(t)
    print(f"  {k:30s}: median {t[len(t) // 2]:.3f} ms (min {t[0]:.3f})")
