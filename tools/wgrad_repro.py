"""Run one convolution's backward twice per size and compare the weight / bias gradients bitwise (diagnostic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.engine import Engine
from vcm_ts_amd.grad import Tape

dev = torch.device("cuda:0")
e = Engine(dev, "fp16x3")
noise_stream = torch.cuda.Stream()
A = torch.randn(2048, 2048, device=dev)
import random
for ks, cin, cout in ((7, 8, 32), (7, 32, 64), (7, 16, 2), (3, 64, 64)):
    for size in (16, 32, 64):
        g = torch.Generator().manual_seed(size)
        N = 2
        w = (torch.randn(cout, cin, ks, ks, generator=g) * 0.2).to(dev).requires_grad_()
        b = (torch.randn(cout, generator=g) * 0.1).to(dev).requires_grad_()
        x = torch.randn(N, cin, size, size, generator=g).to(dev)
        dout = torch.randn(N, cout, size, size, generator=g).to(dev)
        res = []
        for rep in range(3):
            with torch.cuda.stream(noise_stream):  # unrelated work sharing the CUs while the gradient kernels run
                for _ in range(random.randint(1, 6)):
                    A2 = A @ A
            tape = Tape(e)
            e.tape = tape
            xv = e.from_nchw(x, e.buf(f"x{rep}", N, size, size, cin))
            pk = e.pack(("t", ks, cin, cout, size), w, b, (cin,), False)
            out = e.buf(f"o{rep}", N, size, size, cout)
            e.conv(pk, [xv], out, out_slope=0.0)
            e.tape = None
            e.from_nchw(dout, tape.grad(out))
            tape.backward()
            torch.cuda.synchronize()
            res.append((tape.pgrads[id(w)].clone(), tape.pgrads[id(b)].clone()))
        same = all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])
        d = max(float((res[0][0] - r[0]).abs().max()) for r in res[1:])
        if not same:
            dd = (res[0][0] - res[1][0]).abs()
            nz = dd.nonzero()
            print("   differing elements:", nz.shape[0], "of", dd.numel(), "first:", nz[:6].tolist(), "bias same:", torch.equal(res[0][1], res[1][1]))
        print(f"k{ks} {cin}->{cout} {size}x{size}: identical={same} max|dw diff|={d:.3e} |dw|max={float(res[0][0].abs().max()):.3e}")
