"""Does the LDS-reduction version of warp_bwd's flow gradient (tools/probes/warp_bwd_lds_probe.hip, variant 0) give the
same result every time while weight-gradient kernels run on a second stream?  (variant 1: the shuffle version; 2: LDS arrays behind 4 KiB of padding; 3: arrays swapped; 4: volatile LDS accesses;
5: v_mov copies before the LDS write.)
usage: python3 tools/warp_bwd_lds_probe.py [reps]      PROBE_LIB=libwarp_bwd_lds_probe_noslp.so: the same kernels built with
-fno-slp-vectorize (no packed-FP32 instructions): never differs"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd import lib
from vcm_ts_amd.engine import Engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
e = Engine(dev, "fp16x3")
P = C.CDLL(os.path.join(ROOT, "tools", "probes", os.environ.get("PROBE_LIB", "libwarp_bwd_lds_probe.so")))
vp, i32 = C.c_void_p, C.c_int32
P.probe_warp_dflow.argtypes = [i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp]
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()
wx = torch.randn(4, 64, 64, 64, generator=g).to(dev)
wd = torch.randn(4, 64, 64, 64, generator=g).to(dev)
dw = torch.empty(64, 64, 3, 3, device=dev)
sc = torch.empty(24 * 1024 * 1024, device=dev)


def wgrad(prec):
    a = lib.WgradArgs()
    a.x, a.x_cs, a.C = wx.data_ptr(), 64, 64
    a.dpre, a.dpre_cs, a.zs, a.Hd, a.Wd = wd.data_ptr(), 64, 1, 64, 64
    a.N, a.Hin, a.Win, a.Ho, a.Wo, a.Cout, a.ks, a.stride = 4, 64, 64, 64, 64, 64, 3, 1
    a.dw, a.Cin_total, a.cin_offset = dw.data_ptr(), 64, 0
    a.scratch, a.scratch_floats, a.overwrite, a.precision = sc.data_ptr(), sc.numel(), 1, prec
    lib.check(e.L.dcvc_conv_wgrad(C.byref(a), C.c_void_p(side.cuda_stream)), "wgrad")


cw = torch.nn.Parameter((torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev))
cb = torch.nn.Parameter(torch.zeros(64, device=dev))
cpk = e.pack(("probe",), cw, cb, (64,), False)
cx = e.buf("probe/x", 2, 256, 256, 64); cx.base.normal_()
co = e.buf("probe/o", 2, 256, 256, 64)


def conv_load():   # the fp16x3 forward convolution (fp16 MFMA, 64 KB of LDS) on the side stream
    with torch.cuda.stream(side):
        e.conv(cpk, [cx], co, out_slope=0.01)


for (N, H, W, Cc) in ((4, 128, 128, 64), (4, 256, 256, 64)):
    cs = (Cc + 3) // 4 * 4
    src = torch.randn(N, H, W, cs, generator=g).to(dev)
    flow = (torch.randn(N, H, W, 4, generator=g) * 0.02).to(dev)
    dout = (torch.randn(N, H, W, cs, generator=g) * 1e-3).to(dev)
    fix = torch.zeros(N * H * W * Cc, dtype=torch.int64, device=dev)
    for variant in (0, 2, 3, 4, 5, 1):
        for load, tag in ((1, "bf16 wgrad alongside"), (1, "bf16 wgrad alongside, no atomics"), (0, "fp32 wgrad alongside"), (2, "fp16x3 conv alongside"), (None, "alone")):
            outs, bad = None, 0
            for rep in range(reps):
                dflow = torch.zeros(N, H, W, 4, device=dev)
                fix.zero_()
                if load is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    for _ in range(6):
                        conv_load() if load == 2 else wgrad(load)
                P.probe_warp_dflow(variant, src.data_ptr(), cs, flow.data_ptr(), 4, dout.data_ptr(), cs, dflow.data_ptr(), 4,
                                   N, H, W, Cc, None if tag.endswith('no atomics') else fix.data_ptr(), e.stream())
                torch.cuda.synchronize()
                if outs is None:
                    outs = dflow
                elif not torch.equal(outs, dflow):
                    bad += 1
                    last = ((outs != dflow).sum(dim=(0, 1, 2)).tolist(), float((outs - dflow).abs().max()))
            print(f"variant {variant} {N}x{H}x{W}x{Cc} {tag}: {bad} of {reps - 1} repeats differ" + (f" (per channel {last[0]}, max diff {last[1]:.2e})" if bad else ""))
