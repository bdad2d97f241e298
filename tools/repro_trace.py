"""Clone every gradient buffer at the END of the backward pass of identical runs; list the buffers that differ (diagnostic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.synthetic import frames
from vcm_ts_amd import grad as G

dev = torch.device("cuda:0")
m = DMC(precision="fp16x3").to(dev).train()
for p in m.parameters():
    p.requires_grad_(True)
N, size = 2, 128
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
    # name the buffers by the forward op that produced them
    names = {}
    for n, op in enumerate(self.ops):
        for a in op[1:]:
            vs = a if isinstance(a, (list, tuple)) else [a]
            for v in vs:
                if isinstance(v, G.View):
                    names.setdefault(v.base.data_ptr(), f"{n}:{op[0]}:{getattr(op[1], 'key', '')}"[:90])
    snaps.append({names.get(k, str(k)): t.clone() for k, t in self.gbufs.items()})


G.Tape.backward = traced_backward


def run():
    dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    m.zero_grad(set_to_none=True)
    out = m.forward_one_frame(x1, dpb, 1.0, 1.0)
    loss = torch.mean(out["bpp"] + 256.0 * out["mse"] + out["me_mse"])
    loss.backward()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


gr = [run() for _ in range(4)]
torch.cuda.synchronize()
for j in range(1, 4):
    badp = [k for k in gr[0] if not torch.equal(gr[0][k], gr[j][k])]
    bad = [(k, float((snaps[0][k] - snaps[j][k]).abs().max()), float(snaps[0][k].abs().max())) for k in snaps[0] if k in snaps[j] and not torch.equal(snaps[0][k], snaps[j][k])]
    print(f"run 0 vs {j}: {len(badp)} parameter gradients differ; {len(bad)} of {len(snaps[0])} gradient buffers differ")
    for b in bad[:40]:
        print("    ", b)

for key in snaps[1]:
    if key.startswith("19:conv") or key.startswith("20:conv") or key.startswith("14:up2") or key.startswith("21:up2"):
        a, b = snaps[1][key], snaps[2][key]
        d = (a - b).abs()
        nz = d.nonzero().flatten()
        print(key[:60], "numel", a.numel(), "differing", nz.numel(), "first idx", nz[:8].tolist(), "last", nz[-4:].tolist() if nz.numel() else [])
        if nz.numel():
            cs = 32 if "conv4" in key else (4 if "conv5" in key else 8)
            pix = (nz // cs)
            ch = (nz % cs)
            Wd = 64
            ys, xs = (pix // Wd) % 64, pix % Wd
            print("     channels:", sorted(set(ch.tolist()))[:40], " y range", int(ys.min()), int(ys.max()), " x range", int(xs.min()), int(xs.max()),
                  " images", sorted(set((pix // (64 * 64)).tolist())))
