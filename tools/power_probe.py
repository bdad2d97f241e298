"""Board power / clocks while the dominant convolution runs back to back on RANDOM and on ALL-ZERO operands (developer
probe; rocm-smi sampled every 0.5 s from a thread).  Companion of tools/conv_data_probe.py: that one says the kernel is
24 % faster on zeros, this one says why -- package power against the board's cap and the shader clock the chip holds.
usage: power_probe.py [seconds per case] [k32|mfma]"""
import json, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.engine import Engine
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
use_k32 = (sys.argv[2] if len(sys.argv) > 2 else "k32") == "k32"
H, W, cin, cout = 1088, 1920, 64, 64
e = Engine("cuda:0", precision="fp16x3")
e.use_k32 = use_k32
cases = {}
for tag in ("rand", "zero"):
    x = e.buf(f"x{tag}", 1, H, W, cin); r = e.buf(f"r{tag}", 1, H, W, cout); o = e.buf(f"o{tag}", 1, H, W, cout)
    if tag == "rand":
        x.base.normal_(); r.base.normal_(); w = torch.randn(cout, cin, 3, 3) * 0.05
    else:
        x.base.zero_(); r.base.zero_(); w = torch.zeros(cout, cin, 3, 3)
    pk = e.pack((tag,), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(torch.zeros(cout).cuda()), (cin,), False)
    cases[tag] = (lambda x=x, o=o, pk=pk: e._conv_f32(pk, [x], o, 1, None, 0.01, None, None, None))
print(f"# {'conv_k32<3,4,8>' if use_k32 else 'conv_mfma<3,1,2,2,true>'} (64,)->64 3x3 1088x1920 fp16x3, back-to-back launches, {secs:.0f} s per case")
for tag, fn in cases.items():
    samples, stop = [], False
    def poll():
        while not stop:
            try:
                samples.append(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout.strip())
            except Exception as ex:
                samples.append(repr(ex))
            time.sleep(0.5)
    for _ in range(50): fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(200): fn()
        torch.cuda.synchronize(); n += 200
    dt = time.time() - t0
    stop = True; th.join()
    print(f"{tag}: {dt / n * 1e3:.3f} ms per launch over {n} launches")
    for s in samples[2:8]:
        try:
            d = json.loads(s); c = d[sorted(d)[0]]
            print("   ", {k: v for k, v in c.items() if "ower" in k or "sclk" in k})
        except Exception:
            print("   ", s[:200])
