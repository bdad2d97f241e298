"""Generate tests/golden/train_64.npz and train_256_b4.npz by running the REFERENCE's training-mode forward and
torch.autograd backward (build container only; see tests/golden/make_golden.py for the import rules).

Two consecutive P pictures of a 64x64 batch-2 clip: the first after an "I picture" (DPB holds
only ref_frame), the second with the full (detached) DPB -- the reference's `single` training
recursion (core/model/dcvc_hem.py:189-196).  Stored: the uniform draws add_noise made (so that
a checker can replay them), every scalar output, the loss and, per parameter, the gradient's
L2 norm and its first 8 values, the ten largest gradients in full (unit vectors, fp16); gradients of the per-sample
q-scales in full; since round 4 also the integers the reference's roundings produced (hyper latents, dual-prior
residuals) for forced-symbol replay.

train_256_b4 is BASELINE configs[2]'s shape: batch 4 of 256x256 pictures, one rate point per sample (the
model's first four q-scales, lambdas 85 / 170 / 380 / 840 as core/config/defaults.py).

    python tests/golden/make_golden_train.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from refimport import load  # noqa: E402
from vcm_ts_amd.params import dmc_spec, seeded_state_dict  # noqa: E402
from vcm_ts_amd.synthetic import frames  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LAMBDA, ME_WEIGHT = 50.0, 10.0   # loss = mean(bpp + LAMBDA * mse + ME_WEIGHT * me_mse)


def main(N=2, size=64, out_name="train_64", lambdas=None):
    DMC, _ = load(with_cxx=False)
    net = DMC(anchor_num=4)
    net.load_state_dict(seeded_state_dict(dmc_spec()))
    net.train()
    fr = frames(3, N * 3, size, size)
    x0, x1, x2 = (torch.from_numpy(fr[k * N:(k + 1) * N]) for k in range(3))
    if lambdas is None:
        q_mv = torch.tensor([1.0, 0.8]).view(N, 1, 1, 1)
        q_y = torch.tensor([1.2, 0.9]).view(N, 1, 1, 1)
        lam = torch.full((N,), LAMBDA)
    else:  # one rate point per sample (core/data/__init__.py:75)
        q_mv = net.mv_y_q_scale[:N].detach().clone().view(N, 1, 1, 1)
        q_y = net.y_q_scale[:N].detach().clone().view(N, 1, 1, 1)
        lam = torch.tensor(lambdas, dtype=torch.float32)
    draws = []
    orig = net.add_noise

    def add_noise(x):
        out = orig(x)
        draws.append((out - x).detach().clone())
        return out

    net.add_noise = add_noise
    # round 4 (forced-symbol replay): the integers the reference's roundings produced in this very forward -- the two
    # rounded hyper latents (quant(), video_model.py:488,514) and the rounded dual-prior residuals y_q
    # (forward_dual_prior's second result, :499,526) -- so that a checker can feed them to its own forward and compare
    # gradients without a value that sits on a rounding tie falling the other way
    rounded = {}
    net.mv_hyper_prior_encoder.register_forward_hook(lambda m, i, o: rounded.__setitem__("mv_z", torch.round(o.detach())))
    net.contextual_hyper_prior_encoder.register_forward_hook(lambda m, i, o: rounded.__setitem__("z", torch.round(o.detach())))
    orig_dual = net.forward_dual_prior

    def dual(y, means, scales, qs, prior, write=False):
        res = orig_dual(y, means, scales, qs, prior, write=write)
        rounded["mv" if prior is net.mv_y_spatial_prior else "y"] = res[1].detach().clone()
        return res

    net.forward_dual_prior = dual
    fx = {"meta": np.array([N, size, LAMBDA, ME_WEIGHT], np.float64), "names": np.array(list(dmc_spec().keys())),
          "lambdas": lam.numpy().astype(np.float64), "q_mv": q_mv.reshape(-1).numpy().astype(np.float64),
          "q_y": q_y.reshape(-1).numpy().astype(np.float64)}
    dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for step, x in enumerate((x1, x2)):
        torch.manual_seed(100 + step)
        draws.clear()
        net.zero_grad(set_to_none=True)
        qm, qy = q_mv.clone().requires_grad_(), q_y.clone().requires_grad_()
        out = net.forward_one_frame(x, dpb, qm, qy)
        loss = torch.mean(out["bpp"] + lam * out["mse"] + ME_WEIGHT * out["me_mse"])
        loss.backward()
        p = f"s{step}_"
        for key, t in zip(("y", "mv_y", "z", "mv_z"), draws):   # order of the calls at video_model.py:547-550
            fx[p + "noise_" + key] = t.numpy().astype(np.float32)
        for key in ("bpp_y", "bpp_z", "bpp_mv_y", "bpp_mv_z", "bpp", "mse", "me_mse"):
            fx[p + key] = out[key].detach().numpy().astype(np.float64)
        fx[p + "loss"] = np.float64(loss.item())
        for key, t in rounded.items():
            a = t.numpy()
            assert np.array_equal(a, np.rint(a)) and np.abs(a).max() < 32768, key
            fx[p + "rounded_" + key] = a.astype(np.int8 if np.abs(a).max() < 128 else np.int16)
        fx[p + "dq_mv"] = qm.grad.numpy().astype(np.float64)
        fx[p + "dq_y"] = qy.grad.numpy().astype(np.float64)
        norms, heads = [], []
        for name, prm in net.named_parameters():
            g = prm.grad
            if g is None:
                norms.append(-1.0)
                heads.append(np.zeros(8, np.float32))
            else:
                norms.append(g.double().norm().item())
                h = np.zeros(8, np.float32)
                flat = g.reshape(-1)[:8].numpy()
                h[:flat.size] = flat
                heads.append(h)
        fx[p + "grad_names"] = np.array([n for n, _ in net.named_parameters()])
        fx[p + "grad_norm"] = np.array(norms, np.float64)
        fx[p + "grad_head"] = np.stack(heads)
        # directions, not only lengths: the ten tensors that carry most of the gradient, in full, as unit
        # vectors in fp16 (the norm is stored above; fp16 keeps a cosine to ~1e-7)
        named = list(net.named_parameters())
        top = sorted(range(len(named)), key=lambda j: -norms[j])[:10]
        fx[p + "grad_full_index"] = np.array(top, np.int64)
        for j in top:
            g = named[j][1].grad.double()
            fx[p + f"grad_full_{j}"] = (g / g.norm()).reshape(-1).numpy().astype(np.float16)
        dpb = {k: v.detach() for k, v in out["dpb"].items()}
        print(f"step {step}: loss {loss.item():.6f}, {sum(n >= 0 for n in norms)} parameter gradients")
    np.savez_compressed(os.path.join(OUT, out_name + ".npz"), **fx)


def main_intra(N=2, size=64, out_name="train_intra_64"):
    """IntraNoAR in training mode (round 4): the reference's trainers run it under no_grad (core/model/dcvc_hem.py:164-167),
    but image_model.py:54-106 is differentiable; loss = mean(bpp + LAMBDA * mse), one q-scale per sample.  Same contents as
    the DMC fixtures: the two add_noise draws (y_res, z), scalars, per-parameter gradient norms and first values, dq."""
    from vcm_ts_amd.params import intra_spec

    _, IntraNoAR = load(with_cxx=False)
    net = IntraNoAR()
    net.load_state_dict(seeded_state_dict(intra_spec()))
    net.train()
    x = torch.from_numpy(frames(9, N, size, size))
    q0 = torch.tensor([1.0, 0.8][:N]).view(N, 1, 1, 1)
    draws, orig = [], net.add_noise

    def add_noise(t):
        out = orig(t)
        draws.append((out - t).detach().clone())
        return out

    net.add_noise = add_noise
    torch.manual_seed(200)
    q = q0.clone().requires_grad_()
    out = net(x, q)
    loss = torch.mean(out["bpp"] + LAMBDA * out["mse"])
    loss.backward()
    fx = {"meta": np.array([N, size, LAMBDA], np.float64), "q": q0.reshape(-1).numpy().astype(np.float64),
          "noise_y": draws[0].numpy().astype(np.float32), "noise_z": draws[1].numpy().astype(np.float32),
          "loss": np.float64(loss.item()), "dq": q.grad.numpy().astype(np.float64), "bit": np.float64(out["bit"])}
    for key in ("bpp", "bpp_y", "bpp_z", "mse"):
        fx[key] = out[key].detach().numpy().astype(np.float64)
    names, norms, heads = [], [], []
    for name, prm in net.named_parameters():
        names.append(name)
        g = prm.grad
        norms.append(-1.0 if g is None else g.double().norm().item())
        h = np.zeros(8, np.float32)
        if g is not None:
            flat = g.reshape(-1)[:8].numpy()
            h[:flat.size] = flat
        heads.append(h)
    fx["grad_names"], fx["grad_norm"], fx["grad_head"] = np.array(names), np.array(norms, np.float64), np.stack(heads)
    named = list(net.named_parameters())
    top = sorted(range(len(named)), key=lambda j: -norms[j])[:4]  # (192 x 192 x 9 filters: four in full are enough)
    fx["grad_full_index"] = np.array(top, np.int64)
    for j in top:
        g = named[j][1].grad.double()
        fx[f"grad_full_{j}"] = (g / g.norm()).reshape(-1).numpy().astype(np.float16)
    np.savez_compressed(os.path.join(OUT, out_name + ".npz"), **fx)
    print(f"{out_name}: loss {loss.item():.6f}, {sum(n >= 0 for n in norms)} parameter gradients")


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "intra":
        main_intra()
    else:
        main()
        main(4, 256, "train_256_b4", (85.0, 170.0, 380.0, 840.0))
        main_intra()
