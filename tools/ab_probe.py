"""Same-box A/B of builds of libdcvc_hip.so on one convolution layer: every library is loaded in its own child process
(DCVC_HIP_LIB), rounds are interleaved across the children so that clock / thermal drift hits all of them alike.
usage: ab_probe.py lib1.so[:ENV=VAL[:ENV2=VAL2]] lib2.so ... [-- cin cout ks [H W]]   ("-" = the product library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("AB_CHILD"):
    import torch
    sys.path.insert(0, ROOT)
    from vcm_ts_amd.engine import Engine
    cin, cout, ks, H, W = (int(v) for v in os.environ["AB_CHILD"].split(","))
    e = Engine("cuda:0", precision="fp16x3")
    x = e.buf("x", 1, H, W, cin); r = e.buf("r", 1, H, W, cout); o = e.buf("o", 1, H, W, cout)
    x.base.normal_(); r.base.normal_()
    pk = e.pack(("w",), torch.nn.Parameter((torch.randn(cout, cin, ks, ks) * 0.05).cuda()), torch.nn.Parameter(torch.zeros(cout).cuda()), (cin,), False)
    def run(res):
        e._conv_f32(pk, [x], o, 1, None, 0.01, r if res else None, None, None)
    for _ in range(30): run(False); run(True)
    torch.cuda.synchronize()
    print("ready", flush=True)
    for line in sys.stdin:
        out = []
        for res in (False, True):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5): run(res)
            e0.record()
            for _ in range(30): run(res)
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 30)
        print(f"{out[0]:.4f} {out[1]:.4f}", flush=True)
    sys.exit(0)
args = sys.argv[1:]
shape = [64, 64, 3, 1088, 1920]
if "--" in args:
    k = args.index("--"); extra = [int(v) for v in args[k + 1:]]; args = args[:k]
    shape[:len(extra)] = extra
kids = []
for spec in args:
    lib, *sets = spec.split(":")
    env = dict(os.environ, AB_CHILD=",".join(map(str, shape)))
    if lib != "-":
        env["DCVC_HIP_LIB"] = lib
    for kv in sets:
        k_, v_ = kv.split("=", 1)
        env[k_] = v_
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    assert p.stdout.readline().strip() == "ready", spec
    kids.append(p)
times = [[] for _ in kids]
for rnd in range(9):
    for i, p in enumerate(kids):
        p.stdin.write("go\n"); p.stdin.flush()
        times[i].append([float(v) for v in p.stdout.readline().split()])
for p in kids:
    p.stdin.close(); p.wait()
print(f"({shape[0]},)->{shape[1]} k{shape[2]} {shape[3]}x{shape[4]} fp16x3 random data; median of 9 interleaved rounds x 30 launches, ms: no residual / residual")
for lib, t in zip(args, times):
    a = sorted(v[0] for v in t)[4]; b = sorted(v[1] for v in t)[4]
    print(f"  {os.path.basename(lib):28s} {a:.4f} / {b:.4f}")
