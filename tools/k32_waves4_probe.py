"""conv_k32's 64-output-channel 3x3 kernel as 8-wave (8-row tile, two workgroups per CU: the product) and as 4-wave
(16-row tile, one workgroup per CU: round-4 experiment) workgroups on one layer: results compared bit for bit, times interleaved.
usage: k32_waves_probe.py [cin cout [H W]]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd import lib
from vcm_ts_amd.engine import Engine
argv = [int(v) for v in sys.argv[1:]]
cin, cout = argv[:2] if len(argv) >= 2 else (64, 64)
H, W = argv[2:4] if len(argv) >= 4 else (1088, 1920)
e = Engine("cuda:0", precision="fp16x3")
x = e.buf("x", 1, H, W, cin); r = e.buf("r", 1, H, W, cout); o = e.buf("o", 1, H, W, cout)
x.base.normal_(); r.base.normal_()
pk = e.pack(("w",), torch.nn.Parameter((torch.randn(cout, cin, 3, 3) * 0.05).cuda()), torch.nn.Parameter(torch.randn(cout).cuda()), (cin,), False)
e.k32_everywhere = True
outs = {}
for waves in (8, 4):
    lib.check(e.L.dcvc_conv_k32_set_waves(waves), "set_waves")
    for res in (False, True):
        o.base.fill_(float("nan"))
        e._conv_f32(pk, [x], o, 1, None, 0.01, r if res else None, None, None)
        outs[(waves, res)] = o.base.clone()
for res in (False, True):
    same = torch.equal(outs[(8, res)], outs[(4, res)])
    print(f"residual={res}: 4-wave output bit-identical to the 8-wave output: {same}; finite: {bool(torch.isfinite(outs[(4, res)]).all())}")
times = {(w, res): [] for w in (8, 4) for res in (False, True)}
for rnd in range(9):
    for waves in (8, 4):
        e.L.dcvc_conv_k32_set_waves(waves)
        for res in (False, True):
            for _ in range(5):
                e._conv_f32(pk, [x], o, 1, None, 0.01, r if res else None, None, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                e._conv_f32(pk, [x], o, 1, None, 0.01, r if res else None, None, None)
            e1.record(); torch.cuda.synchronize()
            times[(waves, res)].append(e0.elapsed_time(e1) / 30)
e.L.dcvc_conv_k32_set_waves(8)
print(f"({cin},)->{cout} k3 {H}x{W} fp16x3 random data; median of 9 interleaved rounds x 30 launches, ms: no residual / residual")
for waves in (8, 4):
    print(f"  {waves:2d} waves per workgroup: {sorted(times[(waves, False)])[4]:.4f} / {sorted(times[(waves, True)])[4]:.4f}")
