"""One steady-state chunk (3 filter rows) of the persistent conv_k32p kernel, stamped (tools/probes/conv_k32_stamps.hip):
where a workgroup's cycles go between the phase boundaries.  64->64 3x3 at 1088x1920 has 2 chunks per tile; the stamped
chunk is the first chunk of each workgroup's third tile.   usage: conv_k32p_stamps.py [rand|zero] [res]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DCVC_K32_PERSISTENT"] = "2"
from vcm_ts_amd import lib
from vcm_ts_amd.engine import Engine
data = sys.argv[1] if len(sys.argv) > 1 else "rand"
res = len(sys.argv) > 2
cin = cout = 64
H, W = 1088, 1920
e = Engine("cuda:0", precision="fp16x3")
P = C.CDLL(os.path.join(ROOT, "tools", "probes", "libconv_k32_stamps.so"))
P.dcvc_conv2d_k32.argtypes = [C.POINTER(lib.ConvArgs), C.c_void_p]; P.dcvc_conv2d_k32.restype = C.c_int
P.k32_stamps_set.argtypes = [C.c_void_p]
x = e.buf("x", 1, H, W, cin); r = e.buf("r", 1, H, W, cout); o = e.buf("o", 1, H, W, cout)
if data == "rand":
    x.base.normal_(); r.base.normal_(); w = torch.randn(cout, cin, 3, 3) * 0.05
else:
    x.base.zero_(); r.base.zero_(); w = torch.zeros(cout, cin, 3, 3)
pk = e.pack(("p",), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(torch.zeros(cout).cuda()), (cin,), False)
nwg = 256
stamps = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
assert P.k32_stamps_set(stamps.data_ptr()) == 0
e.L.dcvc_conv2d_k32 = P.dcvc_conv2d_k32
run = lambda: e._conv_f32(pk, [x], o, 1, None, 0.01, r if res else None, None, None)
for _ in range(200): run()
torch.cuda.synchronize(); stamps.zero_(); run(); torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nwg, 64).astype(np.float64)
mhz = 100.0 * (s[:, 10] - s[:, 0]) / (s[:, 61] - s[:, 62])
print(f"(64,)->64 3x3 {H}x{W} {data} data{' + residual' if res else ''}, persistent kernel; in-kernel clock {np.median(mhz):.0f} MHz")
for nm, i0, i1 in (("filter row 0 step (incl. next patch request)", 0, 2), ("barrier", 2, 3), ("filter row 1 step (6 -> 3 patch quads converted)", 3, 5),
                   ("barrier", 5, 6), ("filter row 2 step", 6, 8), ("epilogue branch (only after an item's last chunk)", 8, 9), ("barrier", 9, 10)):
    d = s[:, i1] - s[:, i0]
    print(f"  {nm:60s} median {np.median(d):7.0f}  mean {d.mean():7.0f} cycles")
print("  inside filter row 0:")
for nm, i0, i1 in (("store filter row 1 to LDS (waits for its loads)", 0, 11), ("request filter row 2", 11, 12), ("request next patch (6 loads)", 12, 13),
                   ("tap 0: 12 fragment reads + 24 MFMAs", 13, 14), ("tap 1", 14, 15), ("tap 2", 15, 16), ("to the end of the step", 16, 2)):
    d = s[:, i1] - s[:, i0]
    print(f"    {nm:58s} median {np.median(d):7.0f}  mean {d.mean():7.0f} cycles")
d = s[:, 10] - s[:, 0]
print(f"  chunk total {np.median(d):.0f} cycles (each stamp costs ~300 of them); MFMA issue of a SIMD's two waves 3 x 2304 = 6912")
