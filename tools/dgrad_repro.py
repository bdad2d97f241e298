"""Data-gradient convolution (2->16, 7x7, in place) repeated while weight-gradient kernels run on a second stream:
are its results identical every time?  (diagnostic for a concurrency-dependent difference)"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd import lib
from vcm_ts_amd.engine import Engine, View

dev = torch.device("cuda:0")
e = Engine(dev, "fp16x3")
g = torch.Generator().manual_seed(1)
N, H, W = 2, 64, 64
side = torch.cuda.Stream()
# weight-gradient load for the side stream: 64 -> 32 7x7 at 128x128
wx = torch.randn(N, 128, 128, 64, generator=g).to(dev)
wd = torch.randn(N, 128, 128, 32, generator=g).to(dev)
dw = torch.empty(32, 64, 7, 7, device=dev)
sc = torch.empty(24 * 1024 * 1024, device=dev)


def wgrad(prec):
    a = lib.WgradArgs()
    a.x, a.x_cs, a.C = wx.data_ptr(), 64, 64
    a.dpre, a.dpre_cs, a.zs, a.Hd, a.Wd = wd.data_ptr(), 32, 1, 128, 128
    a.N, a.Hin, a.Win, a.Ho, a.Wo, a.Cout, a.ks, a.stride = N, 128, 128, 128, 128, 32, 7, 1
    a.dw, a.Cin_total, a.cin_offset = dw.data_ptr(), 64, 0
    a.scratch, a.scratch_floats, a.overwrite, a.precision = sc.data_ptr(), sc.numel(), 1, prec
    lib.check(e.L.dcvc_conv_wgrad(C.byref(a), C.c_void_p(side.cuda_stream)), "wgrad")


for cin, cout in ((2, 16), (16, 32), (32, 8)):
    w = (torch.randn(cin, cout, 7, 7, generator=g) * 0.1).to(dev)   # forward layer cout_f = cin here: (Cout_f, Cin_f)
    dpre = torch.empty(N, H, W, (cin + 3) // 4 * 4, device=dev)
    dpre.fill_(float("nan"))
    dpre[..., :cin] = torch.randn(N, H, W, cin, generator=g).to(dev)
    src = View(dpre, cin)
    for prec, tag in ((1, "bf16 wgrad alongside"), (0, "fp32 wgrad alongside"), (None, "alone")):
        outs = []
        for rep in range(6):
            pkT = e.pack_dev(("t", cin, cout), w, None, (cout,), False, cin_slice=(0, cout), transposed=True)
            ds = View(torch.zeros(N, H, W, (cout + 3) // 4 * 4, device=dev), cout)
            if prec is not None:
                side.wait_stream(torch.cuda.current_stream())
                for _ in range(3):
                    wgrad(prec)
            e.conv(pkT, [src], ds, res=ds)
            torch.cuda.synchronize()
            outs.append(ds.base.clone())
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        d = max(float((outs[0] - o).abs().max()) for o in outs[1:])
        print(f"dgrad {cin}->{cout} 7x7 {H}x{W}, {tag}: identical={same} max diff {d:.3e} (|out| max {float(outs[0].abs().max()):.3e})")
