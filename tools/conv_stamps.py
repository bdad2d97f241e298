"""Phase timeline of the convolution main loop from in-kernel s_memtime stamps (probe build).
usage: DCVC_HIP_LIB=.../libprobe_STAMP.so python tools/conv_stamps.py cin cout ks precision"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.engine import Engine
cin, cout, ks, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
H, W = 1088, 1920
e = Engine("cuda:0", precision=prec)
x = e.buf("x", 1, H, W, cin); x.base.normal_()
r = e.buf("r", 1, H, W, cout); r.base.normal_()
o = e.buf("o", 1, H, W, cout)
w = torch.nn.Parameter(torch.randn(cout, cin, ks, ks) * 0.05); b = torch.nn.Parameter(torch.zeros(cout))
pk = e.pack(("p",), w, b, (cin,), False)
nblk = 60 * 136
st = torch.zeros(nblk * 64, dtype=torch.int64, device="cuda:0")
assert e.L.dcvc_probe_set_stamps(C.c_void_p(st.data_ptr())) == 0
for _ in range(3): e.conv(pk, [x], o, out_slope=0.01, res=r)
torch.cuda.synchronize(); st.zero_()
e.conv(pk, [x], o, out_slope=0.01, res=r); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(nblk, 64)
t = s[:, 2:].astype(np.float64)
nchunks = (cin + 15) // 16
# stamps: 0 start, then per step: 1 afterB1, 2 afterStore, 3 afterB2, 4 afterMFMA ; then 5 epilogue end
names = ["start"] + sum([[f"c{c}:B1", f"c{c}:store", f"c{c}:B2", f"c{c}:mfma"] for c in range(nchunks)], []) + ["epi"]
n = len(names)
d = np.diff(t[:, :n], axis=1)
t0 = t[:, 0].min()
print(f"{nblk} workgroups; s_memtime counts shader cycles")
real = (s[:, 1] - s[:, 0]).astype(np.float64)  # s_memrealtime ticks (100 MHz) over the workgroup's life
life = t[:, n - 1] - t[:, 0]
clk = life / real * 100.0
print(f"in-kernel shader clock: median {np.median(clk):.0f} MHz (p10 {np.percentile(clk,10):.0f}, p90 {np.percentile(clk,90):.0f})")
print(f"workgroup lifetime: median {np.median(life):.0f} ticks, p10 {np.percentile(life,10):.0f}, p90 {np.percentile(life,90):.0f}")
for k in range(n - 1):
    print(f"  {names[k]:>10s} -> {names[k+1]:<10s} median {np.median(d[:, k]):7.0f}  p90 {np.percentile(d[:, k], 90):7.0f} ticks")
