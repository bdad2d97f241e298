"""Developer diagnostic: gradients of the HIP training path against the CPU oracle's autograd,
per operator and for a whole P picture.  (The pytest versions live in tests/test_gpu_backward.py.)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dcvc_ref as R  # noqa: E402
from vcm_ts_amd.dmc import DMC  # noqa: E402
from vcm_ts_amd.engine import Engine, View  # noqa: E402
from vcm_ts_amd.grad import Tape  # noqa: E402
from vcm_ts_amd.params import dmc_spec, seeded_state_dict  # noqa: E402
from vcm_ts_amd.synthetic import frames  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def conv_case(e, name, seg_C, Cout, ks, stride, H, W, N=2, in_slope=None, out_slope=None, ps=False, res=False, res2=False,
              gate=False, cin_slice=None, seed=0):
    g = torch.Generator().manual_seed(seed)
    Cin = sum(seg_C)
    CinT = Cin if cin_slice is None else cin_slice[2]
    w = (torch.randn(Cout, CinT, ks, ks, generator=g) * 0.2).to(dev).requires_grad_()
    b = (torch.randn(Cout, generator=g) * 0.1).to(dev).requires_grad_()
    xs = [torch.randn(N, c, H, W, generator=g) for c in seg_C]
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    m = 2 if ps else 1
    Cf = Cout // 4 if ps else Cout
    rs = torch.randn(N, Cf, Ho * m, Wo * m, generator=g) if res else None
    rs2 = torch.randn(N, Cf, Ho * m, Wo * m, generator=g) if res2 else None
    gt = torch.rand(N, Cf, generator=g) if gate else None
    dout = torch.randn(N, Cf, Ho * m, Wo * m, generator=g)
    # ---- reference
    xr = [x.clone().requires_grad_() for x in xs]
    wr, br = w.detach().cpu().clone().requires_grad_(), b.detach().cpu().clone().requires_grad_()
    rr = rs.clone().requires_grad_() if res else None
    rr2 = rs2.clone().requires_grad_() if res2 else None
    gr = gt.clone().requires_grad_() if gate else None
    xin = torch.cat(xr, 1)
    if in_slope is not None:
        xin = F.leaky_relu(xin, in_slope)
    ws = wr if cin_slice is None else wr[:, cin_slice[0]:cin_slice[1]]
    y = F.conv2d(xin, ws, br, stride=stride, padding=pad)
    if out_slope is not None:
        y = F.leaky_relu(y, out_slope)
    if ps:
        y = F.pixel_shuffle(y, 2)
    if res:
        y = y + (rr * gr[:, :, None, None] if gate else rr)
    if res2:
        y = rr2 + y
    y.backward(dout)
    # ---- HIP
    tape = Tape(e)
    e.tape = tape
    vs = [e.from_nchw(x.to(dev), e.buf(f"{name}.x{i}", N, H, W, x.shape[1])) for i, x in enumerate(xs)]
    rv = e.from_nchw(rs.to(dev), e.buf(f"{name}.res", N, Ho * m, Wo * m, Cf)) if res else None
    rv2 = e.from_nchw(rs2.to(dev), e.buf(f"{name}.res2", N, Ho * m, Wo * m, Cf)) if res2 else None
    gv = gt.to(dev).contiguous().view(-1) if gate else None
    pk = e.pack((name,), w, b, tuple(seg_C), ps, None if cin_slice is None else cin_slice[:2])
    out = e.buf(f"{name}.out", N, Ho * m, Wo * m, Cf)
    e.conv(pk, vs, out, stride=stride, in_slope=in_slope, out_slope=out_slope, res=rv, gate=gv, res2=rv2)
    e.tape = None
    fwd = rel(e.to_nchw(out), y)
    e.from_nchw(dout.to(dev), tape.grad(out))
    tape.backward()
    errs = {"fwd": fwd, "dw": rel(tape.pgrads[id(w)], wr.grad), "db": rel(tape.pgrads[id(b)], br.grad)}
    for i, v in enumerate(vs):
        errs[f"dx{i}"] = rel(e.to_nchw(tape.grad(v)), xr[i].grad)
    if res:
        errs["dres"] = rel(e.to_nchw(tape.grad(rv)), rr.grad)
    if res2:
        errs["dres2"] = rel(e.to_nchw(tape.grad(rv2)), rr2.grad)
    if gate:
        errs["dgate"] = rel(tape.vec[gv.data_ptr()].view(N, Cf), gr.grad)
    print(f"conv {name:28s} " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)
    return max(errs.values())


def resample_cases(e):
    g = torch.Generator().manual_seed(1)
    N, H, W = 2, 24, 40
    worst = 0.0
    # upstream gradients of very different magnitudes (feature-level gradients of a mean-reduced loss are ~1e-9, rate
    # gradients reach 1e4): the scatter's fixed point scales itself per call
    for C_, mag in ((3, 1.0), (64, 1.0), (64, 1e-9), (16, 3e4)):
        src = torch.randn(N, C_, H, W, generator=g)
        flow = torch.randn(N, 2, H, W, generator=g) * 3
        dout = torch.randn(N, C_, H, W, generator=g) * mag
        sr, fr = src.clone().requires_grad_(), flow.clone().requires_grad_()
        R.warp(sr, fr).backward(dout)
        tape = Tape(e)
        e.tape = tape
        sv = e.from_nchw(src.to(dev), e.buf("w.src", N, H, W, C_))
        fv = e.from_nchw(flow.to(dev), e.buf("w.flow", N, H, W, 2))
        ov = e.warp(sv, fv, e.buf("w.out", N, H, W, C_))
        e.tape = None
        e.from_nchw(dout.to(dev), tape.grad(ov))
        tape.backward()
        a, b = rel(e.to_nchw(tape.grad(sv)), sr.grad), rel(e.to_nchw(tape.grad(fv)), fr.grad)
        worst = max(worst, a, b)
        print(f"warp C={C_} |dout|~{mag:g}: dsrc={a:.1e} dflow={b:.1e}")
    x = torch.randn(N, 2, H, W, generator=g)
    for nm, fn, oshape in (("up2", lambda t: R.up2(t) * 2.0, (N, 2, 2 * H, 2 * W)),
                           ("down2", lambda t: R.down2(t) / 2, (N, 2, H // 2, W // 2)),
                           ("maxpool2", lambda t: F.max_pool2d(t, 2), (N, 2, H // 2, W // 2))):
        dout = torch.randn(*oshape, generator=g)
        xr = x.clone().requires_grad_()
        fn(xr).backward(dout)
        tape = Tape(e)
        e.tape = tape
        xv = e.from_nchw(x.to(dev), e.buf("r.x", N, H, W, 2))
        ov = e.buf("r.o" + nm, oshape[0], oshape[2], oshape[3], 2)
        if nm == "up2":
            e.up2(xv, ov, scale=2.0)
        elif nm == "down2":
            e.down2(xv, ov, scale=0.5)
        else:
            e.maxpool2(xv, ov)
        e.tape = None
        e.from_nchw(dout.to(dev), tape.grad(ov))
        tape.backward()
        a = rel(e.to_nchw(tape.grad(xv)), xr.grad)
        worst = max(worst, a)
        print(f"{nm}: dx={a:.1e}")
    return worst


_ORACLE_STEPS = {}


def _detached(v):
    if torch.is_tensor(v):
        return v.detach()
    if isinstance(v, dict):
        return {k: _detached(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(_detached(x) for x in v)
    return v


def frame_case(size=64, N=2, second=True, lam=50.0, verbose=True, precision="fp32"):
    torch.manual_seed(0)
    w0 = seeded_state_dict(dmc_spec())
    fr = frames(3, N * 3, size, size)
    x0 = torch.from_numpy(fr[0:N])
    x1 = torch.from_numpy(fr[N:2 * N])
    x2 = torch.from_numpy(fr[2 * N:3 * N])
    m = DMC(precision=precision).to(dev).train()
    for p in m.parameters():
        p.requires_grad_(True)
    q_mv = torch.tensor([1.0, 0.8][:N]).view(N, 1, 1, 1)
    q_y = torch.tensor([1.2, 0.9][:N]).view(N, 1, 1, 1)
    dpb_o = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    dpb_g = {"ref_frame": x0.to(dev), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    worst = 0.0
    for step, x in enumerate([x1, x2][: 2 if second else 1]):
        g = torch.Generator().manual_seed(10 + step)
        noise = {"y": torch.rand(N, 96, size // 16, size // 16, generator=g) - 0.5,
                 "mv_y": torch.rand(N, 64, size // 16, size // 16, generator=g) - 0.5,
                 "z": torch.rand(N, 64, size // 64, size // 64, generator=g) - 0.5,
                 "mv_z": torch.rand(N, 64, size // 64, size // 64, generator=g) - 0.5}
        # the oracle's forward + autograd backward of this step: once per (size, N, step, lam) and process -- the tests
        # run this case in both arithmetic modes against the same oracle numbers
        ck = (size, N, step, lam)
        if ck not in _ORACLE_STEPS:
            w = {k: v.clone().requires_grad_() for k, v in w0.items()}
            qm_o, qy_o = q_mv.clone().requires_grad_(), q_y.clone().requires_grad_()
            with R.training_mode():
                ro = R.dmc_forward_one_frame(w, x, dpb_o, qm_o, qy_o, noise=noise)
            loss_o = torch.mean(ro["bpp"] + lam * ro["mse"] + 10.0 * ro["me_mse"])
            loss_o.backward()
            _ORACLE_STEPS[ck] = (w, qm_o, qy_o, _detached(ro), loss_o.detach())
        w, qm_o, qy_o, ro, loss_o = _ORACLE_STEPS[ck]
        m._noise_override = noise
        m.zero_grad(set_to_none=True)
        qm_g, qy_g = q_mv.clone().to(dev).requires_grad_(), q_y.clone().to(dev).requires_grad_()
        rg = m.forward_one_frame(x.to(dev), dpb_g, qm_g, qy_g)
        loss_g = torch.mean(rg["bpp"] + lam * rg["mse"] + 10.0 * rg["me_mse"])
        loss_g.backward()
        print(f"frame {step}: loss oracle {loss_o.item():.6f} hip {loss_g.item():.6f}; "
              + " ".join(f"{k} {rel(rg[k], ro[k]):.1e}" for k in ("bpp_y", "bpp_z", "bpp_mv_y", "bpp_mv_z", "mse", "me_mse")))
        rows = []
        for k, p in m.named_parameters():
            go = w[k].grad
            gg = p.grad
            if go is None and gg is None:
                continue
            if gg is None:
                rows.append((float("inf"), k, 0.0, float(go.norm())))
                continue
            if go is None:
                rows.append((float("inf") if float(gg.norm()) > 0 else 0.0, k, float(gg.norm()), 0.0))
                continue
            rows.append((rel(gg, go), k, float(gg.norm()), float(go.norm())))
        rows.sort(reverse=True)
        tot_o = torch.cat([w[k].grad.reshape(-1) for k, _ in m.named_parameters() if w[k].grad is not None])
        tot_g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).cpu()
                           for k, p in m.named_parameters() if w[k].grad is not None])
        cos = float(F.cosine_similarity(tot_o.double(), tot_g.double(), dim=0))
        print(f"  all parameters: rel {rel(tot_g, tot_o):.2e} cos {cos:.6f}; dq_mv {rel(qm_g.grad, qm_o.grad):.1e} "
              f"dq_y {rel(qy_g.grad, qy_o.grad):.1e}")
        if verbose:
            for r_, k, a, b_ in rows[:25]:
                print(f"    {r_:9.2e}  |hip| {a:10.3e} |ref| {b_:10.3e}  {k}")
            groups = {}
            for r_, k, a, b_ in rows:
                groups.setdefault(k.split(".")[0], []).append(r_)
            print("  worst per module: " + ", ".join(f"{k}:{max(v):.1e}" for k, v in sorted(groups.items())))
        worst = max(worst, rel(tot_g, tot_o))
        dpb_o = {k: v.detach() for k, v in ro["dpb"].items()}
        dpb_g = {k: v.detach() for k, v in rg["dpb"].items()}
    return worst


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    e = Engine(dev, "fp32")
    if what in ("all", "conv"):
        conv_case(e, "k3s1_act_res", (64,), 64, 3, 1, 20, 36, out_slope=0.01, res=True)
        conv_case(e, "k3s1_plain", (64,), 64, 3, 1, 20, 36)
        conv_case(e, "k3s2_act", (64,), 64, 3, 2, 20, 36, out_slope=0.01)
        conv_case(e, "k1s2", (64,), 64, 1, 2, 20, 36)
        conv_case(e, "k1s1_gate", (32, 32), 64, 1, 1, 20, 36, res=True, gate=True)
        conv_case(e, "k7_relu", (8,), 32, 7, 1, 24, 40, out_slope=0.0)
        conv_case(e, "k7_to2_res", (16,), 2, 7, 1, 24, 40, res=True)
        conv_case(e, "k3_ps_act", (64,), 256, 3, 1, 12, 20, out_slope=0.01, ps=True)
        conv_case(e, "k1_ps", (128,), 256, 1, 1, 12, 20, ps=True)
        conv_case(e, "k3_inact_res2", (128,), 64, 3, 1, 12, 20, in_slope=0.1, out_slope=0.1, res=True, res2=True)
        conv_case(e, "k3_seg3", (64, 64, 96), 96, 3, 1, 8, 12, out_slope=0.2)
        conv_case(e, "k3_cinslice", (128,), 192, 3, 1, 8, 12, out_slope=0.2, cin_slice=(0, 128, 192))
        conv_case(e, "k3_3to64", (3,), 64, 3, 1, 20, 36)
        conv_case(e, "k3_67s2", (3, 64), 64, 3, 2, 20, 36)
        conv_case(e, "k3_64to3", (64,), 3, 3, 1, 20, 36)
    if what in ("all", "resample"):
        resample_cases(e)
    if what in ("all", "frame"):
        frame_case()
    if what == "frame16":
        frame_case(precision="fp16x3", verbose=False)
