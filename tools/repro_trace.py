"""Clone every gradient buffer at the END of the backward pass of identical runs and list the ones that differ, in
backward order (diagnostic).  usage: repro_trace.py [N size]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.synthetic import frames
from vcm_ts_amd import grad as G

KEEP = []
if os.environ.get("KEEP_ALL"):   # experiment: no block is returned to the allocator while a run is in flight
    for nm in ("empty", "zeros", "empty_like", "zeros_like", "full_like"):
        def mk(f):
            def g(*a, **k):
                t = f(*a, **k)
                KEEP.append(t)
                return t
            return g
        setattr(torch, nm, mk(getattr(torch, nm)))
dev = torch.device("cuda:0")
m = DMC(precision="fp16x3").to(dev).train()
for p in m.parameters():
    p.requires_grad_(True)
N, size = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 128)
fr = frames(9, N * 3, size, size)
x0, x1, x2 = (torch.from_numpy(fr[k * N:(k + 1) * N]).to(dev) for k in range(3))
g = torch.Generator().manual_seed(3)
m._noise_override = {"y": torch.rand(N, 96, size // 16, size // 16, generator=g) - 0.5,
                     "mv_y": torch.rand(N, 64, size // 16, size // 16, generator=g) - 0.5,
                     "z": torch.rand(N, 64, size // 64, size // 64, generator=g) - 0.5,
                     "mv_z": torch.rand(N, 64, size // 64, size // 64, generator=g) - 0.5}
snaps = []
orig = G.Tape.backward


def traced_backward(self):
    orig(self)
    names = {}
    for n, op in enumerate(self.ops):
        for a in op[1:]:
            vs = a if isinstance(a, (list, tuple)) else [a]
            for v in vs:
                if isinstance(v, G.View):
                    key = getattr(op[1], "key", "")
                    names.setdefault(v.base.data_ptr(), (n, f"{op[0]}:{key[0][1] if key else ''}"))
    snap = {names.get(k, (-1, str(k))): t.clone() for k, t in self.gbufs.items()}
    snap.update({(1000 + i, "dense"): t.clone() for i, t in enumerate(self.dense.values())})
    snap.update({(2000 + i, "vec"): t.clone() for i, t in enumerate(self.vec.values())})
    snaps.append(snap)
    for n, op in enumerate(self.ops):
        if op[0] == "warp":
            gout = self.grad(op[3], create=False)
            if gout is not None:
                o = (gout.ptr - gout.base.data_ptr()) // 4
                snap[(6000 + n, "warp.dout")] = torch.as_strided(gout.base.flatten(), (gout.N * gout.H * gout.W, gout.C), (gout.cs, 1), o).clone()
                snap[(7000 + n, "warp.shape")] = torch.tensor([op[1].N, op[1].H, op[1].W, op[1].C])
            for tag, off, v in (("warp.src", 4000, op[1]), ("warp.flow", 5000, op[2])):
                o = (v.ptr - v.base.data_ptr()) // 4
                snap[(off + n, tag)] = torch.as_strided(v.base.flatten(), (v.N * v.H * v.W, v.C), (v.cs, 1), o).clone()
    global last_ops
    last_ops = list(self.ops)
    self.keep.append(last_ops)


G.Tape.backward = traced_backward


def run():
    dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    m.zero_grad(set_to_none=True)
    out = m.forward_one_frame(x1, dpb, 1.0, 1.0)
    loss = torch.mean(out["bpp"] + 256.0 * out["mse"] + out["me_mse"])
    loss.backward()
    out_g = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    KEEP.clear()
    return out_g


gr = [run() for _ in range(3)]
torch.cuda.synchronize()
for j in (1, 2):
    badp = [k for k in gr[0] if not torch.equal(gr[0][k], gr[j][k])]
    bad = sorted(((k, int((snaps[0][k] != snaps[j][k]).sum()), snaps[0][k].numel(), float((snaps[0][k] - snaps[j][k]).abs().max()),
                  float(snaps[0][k].abs().max())) for k in snaps[0] if k in snaps[j] and not torch.equal(snaps[0][k], snaps[j][k])), reverse=True)
    print(f"run 0 vs {j}: {len(badp)} parameter gradients differ; {len(bad)} of {len(snaps[0])} gradient buffers differ; first in backward order:")
    for b in bad[:14]:
        print("    ", b)

sys.path.insert(0, ROOT)
from oracle import dcvc_ref as R   # developer diagnostic only: the checker's warp as the reference for the flow gradient
wn = next(k[0] - 7000 for k in sorted(snaps[0]) if 7000 <= k[0] < 8000 and snaps[0][k].tolist()[3] == 64 and snaps[0][k].tolist()[1] == size // 4)   # warp(l3, mv3): the only writer of grad(mv3)
Nn, H, W, Cc = snaps[0][(7000 + wn, "warp.shape")].tolist()
src = snaps[0][(4000 + wn, "warp.src")].view(Nn, H, W, Cc).permute(0, 3, 1, 2).cpu().double()
flow = snaps[0][(5000 + wn, "warp.flow")].view(Nn, H, W, 2).permute(0, 3, 1, 2).cpu().double().requires_grad_()
dout = snaps[0][(6000 + wn, "warp.dout")].view(Nn, H, W, Cc).permute(0, 3, 1, 2).cpu().double()
print("dout identical across runs:", torch.equal(snaps[0][(6000 + wn, "warp.dout")], snaps[1][(6000 + wn, "warp.dout")]))
R.warp(src, flow).backward(dout)
want = flow.grad.permute(0, 2, 3, 1).reshape(-1, 2)
key = next(k for k in sorted(snaps[0], reverse=True) if k[0] < 1000 and snaps[0][k].numel() == Nn * H * W * 4)
for j in range(3):
    got = snaps[j][key].view(-1, 4)[:, :2].cpu().double()
    err = (got - want).abs()
    print(f"run {j}: flow gradient of warp op {wn} ({H}x{W}, C={Cc}) vs torch: max err x {float(err[:,0].max()):.3e}, y {float(err[:,1].max()):.3e}; elements off by > 1e-6: x {int((err[:,0] > 1e-6).sum())}, y {int((err[:,1] > 1e-6).sum())}")
bad = ((snaps[0][key].view(-1, 4)[:, 0] != snaps[1][key].view(-1, 4)[:, 0]).nonzero().flatten())[:10]
for p_ in bad.tolist():
    print("   pixel", p_, "want", want[p_].tolist(), "run0", snaps[0][key].view(-1, 4)[p_, :2].tolist(), "run1", snaps[1][key].view(-1, 4)[p_, :2].tolist())
