"""Where a workgroup of conv_k32 spends its cycles: s_memtime stamps at the phase boundaries of every workgroup
(tools/probes/conv_k32_stamps.hip), one launch of a 3x3 layer at 1088x1920 after a warm-up, random or zero operands.
usage: conv_k32_stamps.py [cin cout [rand|zero]]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd import lib
from vcm_ts_amd.engine import Engine
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 64)
data = sys.argv[3] if len(sys.argv) > 3 else "rand"
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
nwg = ((W + 31) // 32) * ((cout + 63) // 64) * ((H + 7) // 8)
stamps = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
assert P.k32_stamps_set(stamps.data_ptr()) == 0
e.L.dcvc_conv2d_k32 = P.dcvc_conv2d_k32   # the engine's next k32 launches go to the stamped build
run = lambda res: e._conv_f32(pk, [x], o, 1, None, 0.01, r if res else None, None, None)
for res in (False, True):
    for _ in range(200): run(res)          # settle the clock
    torch.cuda.synchronize(); stamps.zero_(); run(res); torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nwg, 64).astype(np.float64)
    nst = (cin // 32) * 3
    t0 = s[:, 0]
    life = s[:, 59] - t0
    mhz = 100.0 * life / (s[:, 63] - s[:, 62])
    print(f"\n({cin},)->{cout} 3x3 {H}x{W} {data} data{' + residual' if res else ''}: {nwg} workgroups, {nst} steps each")
    print(f"  workgroup lifetime: median {np.median(life):9.0f} cycles; in-kernel clock median {np.median(mhz):.0f} MHz")
    rows = [("first loads in flight -> first barrier reached (prologue issue)", s[:, 1] - s[:, 0])]
    wait1 = sum(s[:, 2 + 4 * k] - s[:, 1 + 4 * k] for k in range(nst))
    store = sum(s[:, 3 + 4 * k] - s[:, 2 + 4 * k] for k in range(nst))
    wait2 = sum(s[:, 4 + 4 * k] - s[:, 3 + 4 * k] for k in range(nst))
    mfma = sum((s[:, 1 + 4 * (k + 1)] if k + 1 < nst else s[:, 57]) - s[:, 4 + 4 * k] for k in range(nst))
    rows += [("barrier 1 (all waves done with the previous step), all steps", wait1),
             ("LDS stores (wait for loads, convert, write), all steps", store),
             ("   of which step 0 (first loads: exposed latency + convert)", s[:, 3] - s[:, 2]),
             ("barrier 2 (stores visible), all steps", wait2),
             ("load issue + fragment reads + MFMA issue, all steps", mfma),
             ("barrier before the epilogue", s[:, 58] - s[:, 57]),
             ("epilogue (residual loads, transposes, stores issued)", s[:, 59] - s[:, 58])]
    for name, v in rows:
        print(f"  {name:66s} median {np.median(v):8.0f}  mean {v.mean():8.0f}  ({100 * v.mean() / life.mean():5.1f} % of lifetime)")
    print("  per step (median cycles):  barrier1   stores  barrier2   mfma-phase")
    for k in range(nst):
        m_end = s[:, 1 + 4 * (k + 1)] if k + 1 < nst else s[:, 57]
        print(f"    step {k}: {np.median(s[:, 2 + 4 * k] - s[:, 1 + 4 * k]):14.0f} {np.median(s[:, 3 + 4 * k] - s[:, 2 + 4 * k]):8.0f}"
              f" {np.median(s[:, 4 + 4 * k] - s[:, 3 + 4 * k]):9.0f} {np.median(m_end - s[:, 4 + 4 * k]):12.0f}")
    ideal = (cin // 32) * 9 * 48 * 16
    print(f"  MFMA issue cycles of one wave per tile at 16 cycles per v_mfma_f32_16x16x32_f16: {ideal}")
