"""Board power / clocks while one convolution variant runs back to back (developer probe).
usage: power_probe.py variant seconds     variant: f32act | s16 | s16noDMA | s16noEPI | s16mfma | s16mem"""
import ctypes as C, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd import lib
from vcm_ts_amd.engine import Engine, View
variant, secs = sys.argv[1], float(sys.argv[2])
H, W, cin, cout = 1088, 1920, 64, 64
e = Engine("cuda:0", precision="fp16x3")
x = e.buf("x", 1, H, W, cin, cs=64); x.base.normal_()
o = e.buf("o", 1, H, W, cout, cs=64)
w = torch.nn.Parameter((torch.randn(cout, cin, 3, 3) * 0.05).cuda()); b = torch.nn.Parameter(torch.zeros(cout).cuda())
pk = e.pack(("p",), w, b, (cin,), False)
v = e.s16_pack(x)
only = [View(v.base, v.C, 0, geom=(v.N, v.H, v.W, v.cs, v.ptr), fmt="s16")]
o16 = e.buf("o16", 1, H, W, cout, fmt="s16")
P = C.CDLL(os.path.join(ROOT, "tools", "probes", "libconv_s16_probe.so"))
P.dcvc_conv2d_s16_probe.argtypes = [C.POINTER(lib.ConvS16Args), C.c_int, C.c_void_p]
q = e.pack_s16(pk)
a = lib.ConvS16Args()
a.seg[0].ptr, a.seg[0].C, a.seg[0].cs = only[0].ptr, 64, 64
a.nseg, a.N, a.H, a.W = 1, 1, H, W
a.wpack, a.bpack, a.ks, a.Cout, a.Cout_pad = q.w.data_ptr(), q.b.data_ptr(), 3, 64, 64
a.out_act, a.out_slope, a.out16, a.out16_cs = 1, 0.01, o16.ptr, o16.cs
fns = {"f32act": lambda: e._conv_f32(pk, [x], o, 1, None, 0.01, None, None, None),
       "s16": lambda: P.dcvc_conv2d_s16_probe(C.byref(a), 0, e.stream()),
       "s16noDMA": lambda: P.dcvc_conv2d_s16_probe(C.byref(a), 1, e.stream()),
       "s16noEPI": lambda: P.dcvc_conv2d_s16_probe(C.byref(a), 4, e.stream()),
       "s16mfma": lambda: P.dcvc_conv2d_s16_probe(C.byref(a), 5, e.stream()),
       "s16mem": lambda: P.dcvc_conv2d_s16_probe(C.byref(a), 2, e.stream())}
fn = fns[variant]
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip())
        except Exception as ex:
            samples.append(repr(ex))
        time.sleep(0.5)
th = threading.Thread(target=poll); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(200): fn()
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
stop = True; th.join()
print(f"{variant}: {dt / n * 1e3:.3f} ms per launch over {n} launches")
import json
for s in samples[2:8]:
    try:
        d = json.loads(s); c = d[sorted(d)[0]]
        print({k: v for k, v in c.items() if "ower" in k or "sclk" in k or "mclk" in k or "fclk" in k})
    except Exception:
        print(s[:300])
