"""prints a checksum of a k32 conv output for the library in DCVC_HIP_LIB (or the product)"""
import os, sys, hashlib, torch
sys.path.insert(0, os.getcwd())
from vcm_ts_amd.engine import Engine
e = Engine("cuda:0", precision="fp16x3")
e.k32_everywhere = True
torch.manual_seed(1)
for (cin, cout, H, W, slope) in ((64, 64, 136, 200, None), (64, 64, 136, 200, 0.01), (128, 128, 72, 120, 0.2), (32, 64, 50, 70, None)):
    x = e.buf(f"x{cin}{H}", 1, H, W, cin); o = e.buf(f"o{cout}{H}", 1, H, W, cout)
    g = torch.Generator(device="cuda").manual_seed(3)
    x.base.copy_(torch.randn(x.base.shape, generator=g, device="cuda") * torch.logspace(-9, 3, x.base.numel(), device="cuda").view(x.base.shape))
    pk = e.pack((cin, cout, H), torch.nn.Parameter((torch.randn(cout, cin, 3, 3, generator=g, device="cuda") * 0.05)), torch.nn.Parameter(torch.zeros(cout, device="cuda")), (cin,), False)
    e._conv_f32(pk, [x], o, 1, slope, None, None, None, None)
    torch.cuda.synchronize()
    print(cin, cout, H, W, slope, hashlib.sha256(o.base.cpu().numpy().tobytes()).hexdigest()[:16], bool(torch.isfinite(o.base).all()))
