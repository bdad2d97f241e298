"""Time one convolution signature at 1088x1920 with the library named by $DCVC_HIP_LIB.
usage: conv_probe.py cin cout ks precision [H W]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.engine import Engine
cin, cout, ks, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
H, W = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1088, 1920)
e = Engine("cuda:0", precision=prec)
x = e.buf("x", 1, H, W, cin); x.base.normal_()
r = e.buf("r", 1, H, W, cout); r.base.normal_()
o = e.buf("o", 1, H, W, cout)
w = torch.nn.Parameter(torch.randn(cout, cin, ks, ks) * 0.05); b = torch.nn.Parameter(torch.zeros(cout))
pk = e.pack(("p",), w, b, (cin,), False)
for _ in range(3): e.conv(pk, [x], o, out_slope=0.01, res=r)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
ev0.record()
for _ in range(n): e.conv(pk, [x], o, out_slope=0.01, res=r)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / n
fl = 2.0 * H * W * cin * cout * ks * ks
print(f"{os.path.basename(os.environ.get('DCVC_HIP_LIB', 'default')):28s} {cin}->{cout} k{ks} {prec}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
