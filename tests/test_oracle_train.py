"""The oracle's training-mode forward (straight-through round, noisy latents, LowerBound
gradients) and its torch.autograd gradients against fixtures produced by the reference itself
(tests/golden/make_golden_train.py -> tests/golden/train_64.npz).  CPU only."""
import numpy as np
import torch

from oracle import dcvc_ref as R
from tests.util import golden
from vcm_ts_amd.params import dmc_spec, seeded_state_dict
from vcm_ts_amd.synthetic import frames


def run_oracle_steps(fx):
    N, size, lam, me_w = int(fx["meta"][0]), int(fx["meta"][1]), float(fx["meta"][2]), float(fx["meta"][3])
    w0 = seeded_state_dict(dmc_spec())
    fr = frames(3, N * 3, size, size)
    x0, x1, x2 = (torch.from_numpy(fr[k * N:(k + 1) * N]) for k in range(3))
    q_mv = torch.from_numpy(fx["q_mv"]).float().view(N, 1, 1, 1)
    q_y = torch.from_numpy(fx["q_y"]).float().view(N, 1, 1, 1)
    lam = torch.from_numpy(fx["lambdas"]).float()
    dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for step, x in enumerate((x1, x2)):
        p = f"s{step}_"
        noise = {k: torch.from_numpy(fx[p + "noise_" + k]) for k in ("y", "mv_y", "z", "mv_z")}
        w = {k: v.clone().requires_grad_() for k, v in w0.items()}
        qm, qy = q_mv.clone().requires_grad_(), q_y.clone().requires_grad_()
        with R.training_mode():
            out = R.dmc_forward_one_frame(w, x, dpb, qm, qy, noise=noise)
        loss = torch.mean(out["bpp"] + lam * out["mse"] + me_w * out["me_mse"])
        loss.backward()
        yield step, p, out, loss, w, qm, qy
        dpb = {k: v.detach() for k, v in out["dpb"].items()}


import pytest


@pytest.mark.parametrize("name", ["train_64", "train_256_b4"])
def test_training_forward_and_gradients_match_reference(name):
    """train_256_b4 = BASELINE configs[2]'s shape (batch 4 of 256x256, one rate point and lambda per sample)."""
    fx = golden(name)
    for step, p, out, loss, w, qm, qy in run_oracle_steps(fx):
        for key in ("bpp_y", "bpp_z", "bpp_mv_y", "bpp_mv_z", "bpp", "mse", "me_mse"):
            np.testing.assert_allclose(out[key].detach().numpy(), fx[p + key], rtol=3e-5, err_msg=f"{p}{key}")
        assert abs(loss.item() - float(fx[p + "loss"])) <= 3e-5 * abs(float(fx[p + "loss"]))
        np.testing.assert_allclose(qm.grad.numpy(), fx[p + "dq_mv"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(qy.grad.numpy(), fx[p + "dq_y"], rtol=1e-4, atol=1e-7)
        names = [str(n) for n in fx[p + "grad_names"]]
        worst = 0.0
        for i, name in enumerate(names):
            want = float(fx[p + "grad_norm"][i])
            g = w[name].grad
            if want < 0:
                assert g is None or float(g.norm()) == 0.0, name
                continue
            assert g is not None, name
            got = float(g.double().norm())
            assert abs(got - want) <= 1e-4 * want + 1e-9, (name, got, want)
            head = g.reshape(-1)[:8].numpy()
            np.testing.assert_allclose(head, fx[p + "grad_head"][i][: head.size], rtol=0, atol=1e-4 * want + 1e-9,
                                       err_msg=name)
            worst = max(worst, abs(got - want) / (want + 1e-30))
        print(f"step {step}: worst relative gradient-norm difference {worst:.2e}")
        for j in fx[p + "grad_full_index"]:  # the ten largest gradients in full: directions (fp16 unit vectors)
            want_u = fx[p + f"grad_full_{int(j)}"].astype(np.float64)
            g = w[names[int(j)]].grad.double().reshape(-1).numpy()
            assert g @ want_u / (np.linalg.norm(g) * np.linalg.norm(want_u)) >= 0.999999, names[int(j)]


def test_intra_training_forward_and_gradients_match_reference():
    """IntraNoAR's training-mode forward (image_model.py:54-106 with self.training: Gaussian likelihood of the noisy
    residual, straight-through round) through the oracle against tests/golden/train_intra_64.npz, produced by running the
    reference's IntraNoAR in .train() mode and torch.autograd (make_golden_train.py intra)."""
    from vcm_ts_amd.params import intra_spec

    fx = golden("train_intra_64")
    N, size, lam = int(fx["meta"][0]), int(fx["meta"][1]), float(fx["meta"][2])
    w = {k: v.clone().requires_grad_() for k, v in seeded_state_dict(intra_spec()).items()}
    x = torch.from_numpy(frames(9, N, size, size))
    q = torch.from_numpy(fx["q"]).float().view(N, 1, 1, 1).requires_grad_()
    noise = {"y": torch.from_numpy(fx["noise_y"]), "z": torch.from_numpy(fx["noise_z"])}
    with R.training_mode():
        out = R.intra_forward(w, x, q, noise=noise)
    loss = torch.mean(out["bpp"] + lam * out["mse"])
    loss.backward()
    for key in ("bpp", "bpp_y", "bpp_z", "mse"):
        np.testing.assert_allclose(out[key].detach().numpy(), fx[key], rtol=3e-5, err_msg=key)
    assert abs(loss.item() - float(fx["loss"])) <= 3e-5 * abs(float(fx["loss"]))
    np.testing.assert_allclose(q.grad.numpy(), fx["dq"], rtol=1e-4, atol=1e-7)
    for i, name in enumerate(str(n) for n in fx["grad_names"]):
        want = float(fx["grad_norm"][i])
        g = w[name].grad
        if want <= 0:
            assert g is None or float(g.norm()) == 0.0, name
            continue
        got = float(g.double().norm())
        assert abs(got - want) <= 1e-4 * want + 1e-9, (name, got, want)
        head = g.reshape(-1)[:8].numpy()
        np.testing.assert_allclose(head, fx["grad_head"][i][: head.size], rtol=0, atol=1e-4 * want + 1e-9, err_msg=name)
    for j in fx["grad_full_index"]:
        name = str(fx["grad_names"][int(j)])
        u = fx[f"grad_full_{int(j)}"].astype(np.float64)
        g = w[name].grad.double().reshape(-1).numpy()
        assert float(g @ u / (np.linalg.norm(g) * np.linalg.norm(u))) >= 0.999999, name
