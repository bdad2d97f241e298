"""DCVC_HEM wrapper (vcm_ts_amd/dcvc_hem.py) against the loss assembly restated from
/root/reference/core/model/dcvc_hem.py over the CPU oracle's forward_one_frame: evaluation
results, and training steps (backward + optimiser step inside forward) against the oracle's
autograd gradients."""
import numpy as np
import pytest
import torch

from oracle import dcvc_ref as R
from tests.util import oracle_weights
from vcm_ts_amd.synthetic import frames

pytestmark = pytest.mark.gpu
LAMBDAS = (85.0, 380.0)


@pytest.fixture(scope="module")
def model():
    from vcm_ts_amd.dcvc_hem import build_model, make_cfg

    return build_model(make_cfg(lambdas=LAMBDAS, dist_lambda=1.0, pl_lambda=0.5)).cuda().eval()


def _clip(n=2, t=3, h=64, w=64):
    return torch.from_numpy(np.stack([frames(30 + i, t, h, w) for i in range(n)]))  # (N, T, 3, H, W)


def _oracle_losses(x, p_frames, rate_keys, dist_key):
    """rate / dist / loss per (t_i, p) with dpb starting from the input frame (dcvc_hem.py:181-208)."""
    from vcm_ts_amd.params import dmc_spec, seeded_state_dict

    w = seeded_state_dict(dmc_spec(anchor_num=len(LAMBDAS)))  # the wrapper builds DMC(anchor_num=len(lambdas))
    lam = torch.tensor(LAMBDAS)
    out = []
    for t_i in range(x.shape[1] - p_frames):
        dpb = {"ref_frame": x[:, t_i], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        row = []
        for p in range(p_frames):
            o = R.dmc_forward_one_frame(w, x[:, t_i + 1 + p], dpb, w["mv_y_q_scale"][:2], w["y_q_scale"][:2])
            dpb = o["dpb"]
            rate = sum((o[k] for k in rate_keys), torch.zeros(2))
            row.append((rate, o[dist_key], rate + lam * (o[dist_key] * 1.0), dpb["ref_frame"]))
        out.append(row)
    return out


def test_single_and_cascade_match_oracle(model):
    x = _clip()
    want = _oracle_losses(x, 2, ["bpp_y", "bpp_mv_y"], "mse")
    xg = x.cuda()
    with torch.no_grad():
        s = model("single", xg, xg, "mse", ["bpp_y", "bpp_mv_y"], p_frames=2, perceptual_loss=False, is_train=False)
        c = model("cascade", xg, xg, "mse", ["bpp_y", "bpp_mv_y"], p_frames=2, perceptual_loss=False, is_train=False)
    assert s["rate"].shape == (2, 2) and s["loss_seq"].shape == (2, 1) and s["single_forwards"] == 2
    assert s["input_seqs"].shape == (2, 1, 3, 3, 64, 64) and s["decod_seqs"].shape == (2, 1, 3, 3, 64, 64)
    for p in range(2):
        rate, dist, loss, rec = want[0][p]
        np.testing.assert_allclose(s["rate"][:, p].cpu().numpy(), rate.numpy(), rtol=1e-4)
        np.testing.assert_allclose(s["dist"][:, p].cpu().numpy(), dist.numpy(), rtol=1e-4)
        np.testing.assert_allclose(s["loss"][:, p].cpu().numpy(), loss.numpy(), rtol=1e-4)
        np.testing.assert_allclose(s["decod_seqs"][:, 0, p + 1].cpu().numpy(), rec.numpy(), atol=5e-5)
    np.testing.assert_allclose(s["loss_seq"][:, 0].cpu().numpy(), torch.stack([want[0][0][2], want[0][1][2]], -1).mean(-1).numpy(), rtol=1e-4)
    assert torch.equal(s["input_seqs"][:, 0, 0].cpu(), x[:, 0]) and float(s["p_dist"].abs().max()) == 0.0
    # cascade: means over the p frames of one sub-sequence (:440-452)
    assert c["rate"].shape == (2, 1) and c["single_forwards"] == 1
    np.testing.assert_allclose(c["loss"][:, 0].cpu().numpy(), s["loss"].mean(-1).cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(c["dist"][:, 0].cpu().numpy(), s["dist"].mean(-1).cpu().numpy(), rtol=1e-5)


def test_single_multi_simple_and_perceptual_hook(model):
    x = _clip(t=2).cuda()
    dpb = {"ref_frame": x[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    with torch.no_grad():
        r = model("single_multi", x[:, 1], x[:, 1], "mse", ["bpp"], dpb=dpb, perceptual_loss=False)
    assert set(r) == {"rate", "dist", "p_dist", "loss", "loss_to_opt", "input_seqs", "decod_seqs", "dpb"}
    assert r["loss_to_opt"].shape == () and r["dpb"]["ref_feature"].shape == (2, 64, 64, 64)
    torch.testing.assert_close(r["loss"], r["rate"] + model.lambdas * r["dist"])
    # no rate keys -> lambdas are replaced by ones (:148)
    dpb2 = {"ref_frame": x[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    with torch.no_grad():
        r2 = model("single_multi", x[:, 1], x[:, 1], "mse", [], dpb=dpb2, perceptual_loss=False)
    torch.testing.assert_close(r2["loss"], r2["dist"])
    # pluggable perceptual term
    model.perceptual_loss = lambda target, recon: (target - recon).abs().mean(dim=(1, 2, 3))
    dpb3 = {"ref_frame": x[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    with torch.no_grad():
        r3 = model("single_multi", x[:, 1], x[:, 1], "mse", ["bpp"], dpb=dpb3, perceptual_loss=True)
    torch.testing.assert_close(r3["loss"], r3["rate"] + model.lambdas * (r3["dist"] + 0.5 * r3["p_dist"]))
    model.perceptual_loss = None
    # forward_simple: one sample (rate point) at a time, list of dpbs
    seq = x[:, 1:2]  # (N, 1, 3, H, W): input[i] is (1, 3, H, W)
    dpbs = [{"ref_frame": x[i : i + 1, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None} for i in range(2)]
    with torch.no_grad():
        out = model("forward_simple", seq, dpb=dpbs)
    assert len(out) == 2 and out[0]["ref_y"].shape == (1, 96, 4, 4)
    torch.testing.assert_close(out[1]["ref_frame"], r["dpb"]["ref_frame"][1:2], rtol=1e-4, atol=1e-5)


def _noise(seed, n=2, size=64):
    g = torch.Generator().manual_seed(seed)
    return {"y": torch.rand(n, 96, size // 16, size // 16, generator=g) - 0.5,
            "mv_y": torch.rand(n, 64, size // 16, size // 16, generator=g) - 0.5,
            "z": torch.rand(n, 64, size // 64, size // 64, generator=g) - 0.5,
            "mv_z": torch.rand(n, 64, size // 64, size // 64, generator=g) - 0.5}


def _oracle_step_gradients(x, p_frames, rate_keys, dist_key, noise, cascade):
    """Gradients of the first optimiser step of `single` (one P picture, dcvc_hem.py:189-229) or
    `cascade` (p_frames pictures chained through an un-detached DPB, :411-471) from the oracle."""
    from vcm_ts_amd.params import dmc_spec, seeded_state_dict

    w = {k: v.clone().requires_grad_() for k, v in seeded_state_dict(dmc_spec(anchor_num=len(LAMBDAS))).items()}
    lam = torch.tensor(LAMBDAS)
    dpb = {"ref_frame": x[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    losses = []
    with R.training_mode():
        for p in range(p_frames if cascade else 1):
            o = R.dmc_forward_one_frame(w, x[:, 1 + p], dpb, w["mv_y_q_scale"], w["y_q_scale"], noise=noise)
            dpb = o["dpb"]
            rate = sum((o[k] for k in rate_keys), torch.zeros(2))
            losses.append(rate + lam * o[dist_key])
    torch.stack(losses, -1).mean(-1).mean().backward()
    return {k: v.grad for k, v in w.items() if v.grad is not None}


@pytest.mark.parametrize("method", ["single", "cascade"])
def test_training_step_inside_forward_matches_oracle_gradients(method):
    """is_train=True: the wrapper back-propagates and steps the optimiser inside forward.  With
    plain SGD(lr=1) the parameter change of the first step IS minus the gradient, which must be
    the oracle's autograd gradient of the same loss -- for `cascade` that includes the gradient
    flowing from the second picture into the first through the DPB."""
    from vcm_ts_amd.dcvc_hem import build_model, make_cfg

    model = build_model(make_cfg(lambdas=LAMBDAS, dist_lambda=1.0, pl_lambda=0.0), precision="fp32").cuda().train()
    model.activate_modules_all()
    x = _clip(t=3)
    noise = _noise(5)
    model.dmc._noise_override = noise
    before = {k: v.detach().clone() for k, v in model.dmc.named_parameters()}
    opt = torch.optim.SGD(model.parameters(), lr=1.0)
    p_frames = 2 if method == "cascade" else 1
    clip = x[:, :3] if method == "cascade" else x[:, :2]     # exactly one optimiser step
    r = model(method, clip.cuda(), clip.cuda(), "mse", ["bpp_y", "bpp_mv_y"], p_frames=p_frames, perceptual_loss=False,
              optimizer=opt, is_train=True)
    assert r["single_forwards"] == 1 and torch.isfinite(r["loss"]).all() and not r["loss"].requires_grad
    want = _oracle_step_gradients(x, p_frames, ["bpp_y", "bpp_mv_y"], "mse", noise, cascade=(method == "cascade"))
    num = den = 0.0
    for k, v in model.dmc.named_parameters():
        delta = (before[k] - v.detach()).cpu()
        if k not in want:
            assert float(delta.abs().max()) == 0.0, k
            continue
        g = want[k]
        num += float((delta.double() - g.double()).norm() ** 2)
        den += float(g.double().norm() ** 2)
        assert float((delta - g).norm()) <= 3e-2 * float(g.norm()) + 1e-7, k
    assert (num / den) ** 0.5 < 3e-3
    model.dmc._noise_override = None


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_graphed_training_step_matches_the_eager_one(precision):
    """DMC.graph_training (round 4): a training picture replayed from two captured hipGraphs (forward, reverse pass)
    holds exactly the launches the eager autograd node issues, so four `single_multi` steps with AdamW -- the first
    after an "I picture" (empty DPB: one pair of graphs), three with the full DPB (a second pair, replayed) -- must
    leave BIT-IDENTICAL losses, DPBs and parameters.  The noise draws are shared through static device tensors."""
    from vcm_ts_amd.dcvc_hem import build_model, make_cfg

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    clip = torch.rand(2, 5, 3, 64, 64, generator=g).to(dev)
    noises = [{k: (torch.rand(s, generator=g) - 0.5) for k, s in (("y", (2, 96, 4, 4)), ("mv_y", (2, 64, 4, 4)), ("z", (2, 64, 1, 1)),
                                                               ("mv_z", (2, 64, 1, 1)))} for _ in range(4)]
    runs = {}
    for graphed in (False, True):
        model = build_model(make_cfg(lambdas=LAMBDAS), precision=precision).to(dev).train()
        model.activate_modules_all()
        model.dmc.graph_training = graphed
        static = {k: torch.zeros_like(v, device=dev) for k, v in noises[0].items()}
        model.dmc._noise_override = static
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.99), fused=True)
        dpb = {"ref_frame": clip[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        losses, recs = [], []
        for t in range(1, 5):
            for k, v in noises[t - 1].items():
                static[k].copy_(v)
            opt.zero_grad()
            r = model("single_multi", clip[:, t], clip[:, t], "mse", ["bpp"], perceptual_loss=False, dpb=dpb)
            r["loss_to_opt"].backward()
            opt.step()
            dpb = r["dpb"]
            losses.append(r["loss_to_opt"].detach().clone())
            recs.append(dpb["ref_frame"].detach().clone())
        runs[graphed] = (torch.stack(losses), recs, {k: v.detach().clone() for k, v in model.dmc.named_parameters()})
        if graphed:
            fgs = list(model.dmc._frame_graphs.values())
            assert len(fgs) == 2 and not any(f.busy for f in fgs)   # empty-DPB picture, full-DPB picture
        model.dmc._noise_override = None
    assert torch.equal(runs[True][0], runs[False][0]), (runs[True][0], runs[False][0])
    for a, b in zip(runs[True][1], runs[False][1]):
        assert torch.equal(a, b)
    for k, v in runs[False][2].items():
        assert torch.equal(runs[True][2][k], v), k


def test_cascade_multi_backward_and_checkpoint_round_trip(tmp_path):
    """trainer_multi.py's cascade step (train_multi.py:245-258): the caller back-propagates
    `loss_to_opt` of `cascade_multi` itself; gradients equal the oracle's for the same two chained
    pictures.  Then the usual checkpoint cycle: state_dict -> file -> fresh model gives the same loss."""
    from vcm_ts_amd.dcvc_hem import build_model, make_cfg

    cfg = make_cfg(lambdas=LAMBDAS, dist_lambda=1.0, pl_lambda=0.0)
    model = build_model(cfg, precision="fp32").cuda().train()
    model.activate_modules_all()
    x = _clip(t=3)
    # The noisy-latent rate term has gradient ~1/scale where latent + noise lands within a (tiny)
    # scale's width of +-0.5: for such draws the ORACLE's own gradient moves by percents under a 1e-7
    # input perturbation (seed 9: 4 % on contextual_encoder.conv3.weight), so a comparison is only
    # meaningful for a well-conditioned draw (seed 5: oracle self-sensitivity 7e-5).
    noise = _noise(5)
    model.dmc._noise_override = noise
    xg = x.cuda()
    dpb = {"ref_frame": xg[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    r = model("cascade_multi", xg, xg, "mse", ["bpp_y", "bpp_mv_y"], dpb=dpb, p_frames=2, t_i=0, perceptual_loss=False)
    assert r["loss_to_opt"].requires_grad and r["loss"].shape == (2,)
    r["loss_to_opt"].backward()
    want = _oracle_step_gradients(x, 2, ["bpp_y", "bpp_mv_y"], "mse", noise, cascade=True)
    num = den = 0.0
    for k, v in model.dmc.named_parameters():
        if k in want:
            num += float((v.grad.cpu().double() - want[k].double()).norm() ** 2)
            den += float(want[k].double().norm() ** 2)
        else:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    assert (num / den) ** 0.5 < 3e-3
    path = str(tmp_path / "ckpt.pth")
    torch.save({"model": model.state_dict()}, path)
    again = build_model(cfg, precision="fp32").cuda().train()
    again.load_state_dict(torch.load(path, weights_only=True)["model"])
    again.dmc._noise_override = noise
    dpb2 = {"ref_frame": xg[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    r2 = again("cascade_multi", xg, xg, "mse", ["bpp_y", "bpp_mv_y"], dpb=dpb2, p_frames=2, t_i=0, perceptual_loss=False)
    torch.testing.assert_close(r2["loss"], r["loss"], rtol=1e-6, atol=0)
    model.dmc._noise_override = None


def test_groups_and_dispatch(model):
    x = _clip(t=2).cuda()
    with pytest.raises(ValueError):
        model("nope", x)
    model.activate_modules_inter_dist()
    on = {n for n, p in model.dmc.named_parameters() if p.requires_grad}
    assert on and all(n.startswith(("bit_estimator_z_mv", "mv_", "optic_flow")) for n in on) and "mv_y_q_scale" not in on
    model.activate_modules_inter_dist_rate()
    assert {"mv_y_q_basic", "mv_y_q_scale"} <= {n for n, p in model.dmc.named_parameters() if p.requires_grad}
    model.activate_modules_recon_dist()
    on = {n for n, p in model.dmc.named_parameters() if p.requires_grad}
    assert "y_q_scale" not in on and "contextual_encoder.conv1.weight" in on and "optic_flow.moduleBasic.0.conv1.weight" not in on
    model.activate_modules_recon_dist_rate()
    assert "y_q_scale" in {n for n, p in model.dmc.named_parameters() if p.requires_grad}
    model.activate_modules_all()
    assert all(p.requires_grad for p in model.dmc.parameters())
    assert all(k.startswith("dmc.") for k in model.state_dict())
    for p in model.dmc.parameters():
        p.requires_grad = False
