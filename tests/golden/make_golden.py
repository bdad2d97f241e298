"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Imports /root/reference's DMC / IntraNoAR on CPU (tests/golden/refimport.py), loads the
name-seeded weights of vcm_ts_amd/params.py, feeds the seeded frames of
vcm_ts_amd/synthetic.py and stores small numeric fixtures: scalar outputs, integer
symbol / index planes, CDF tables, statistics and corner crops of the big tensors.
Only these numbers are committed; no reference source travels.

    python tests/golden/make_golden.py            # writes tests/golden/
"""
import os
import struct
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from refimport import load  # noqa: E402
from vcm_ts_amd.params import dmc_spec, intra_spec, seeded_state_dict  # noqa: E402
from vcm_ts_amd.synthetic import frames  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item(), t.sum().item()], np.float64)


def crop(t):
    return t.detach()[..., :8, :8].contiguous().numpy().astype(np.float32)


def build_nets(seed=0, gain=None):
    """seed / gain select another set of name-seeded weights (params.seeded_state_dict): parity must not hinge on
    one particular weight set."""
    DMC, IntraNoAR = load(with_cxx=True)
    torch.manual_seed(0)
    kw = {"seed": seed} if gain is None else {"seed": seed, "gain": gain}
    d = DMC(anchor_num=4)
    d.load_state_dict(seeded_state_dict(dmc_spec(), **kw))
    d.eval()
    i = IntraNoAR()
    i.load_state_dict(seeded_state_dict(intra_spec(), **kw))
    i.eval()
    return d, i


class Tap:
    """Forward hooks on the reference's sub-modules + a recorder around the dual prior."""

    def __init__(self, net, names):
        self.out = {}
        self.dual = []
        for n in names:
            getattr(net, n).register_forward_hook(self._hook(n))
        orig = net.forward_dual_prior

        def rec(y, means, scales, qs, prior, write=False):
            res = orig(y, means, scales, qs, prior, write=write)
            if not write:
                self.dual.append(orig(y, means, scales, qs, prior, write=True))
            return res

        net.forward_dual_prior = rec

    def _hook(self, n):
        def h(mod, inp, out):
            self.out[n] = out

        return h

    def reset(self):
        self.out.clear()
        self.dual.clear()


def planes(net, tap, z_names):
    """int16 symbol planes + int16 index planes in bitstream order."""
    res = {}
    for zi, zn in enumerate(z_names):
        res[f"sym_{zn}"] = torch.round(tap.out[zn]).numpy().astype(np.int16)
    for di, tag in zip(range(len(tap.dual)), ("mv_y", "y") if len(tap.dual) == 2 else ("y",)):
        q0, q1, s0, s1, _ = tap.dual[di]
        res[f"sym_{tag}0"] = q0.numpy().astype(np.int16)
        res[f"sym_{tag}1"] = q1.numpy().astype(np.int16)
        res[f"idx_{tag}0"] = net.gaussian_encoder.build_indexes(s0).numpy().astype(np.int16)
        res[f"idx_{tag}1"] = net.gaussian_encoder.build_indexes(s1).numpy().astype(np.int16)
        res[f"scale_{tag}0"] = s0.numpy().astype(np.float32)
    return res


def sequence_case(d, i, dtap, itap, name, h, w, n_p, seed, batch=1):
    """I-frame + n_p P-frames through the reference's estimate path (forward)."""
    fr = frames(seed, n_p + 1, h, w)
    fx = {}
    if batch == 1:
        xs = [torch.from_numpy(fr[t : t + 1]) for t in range(n_p + 1)]
        iq, mvq, yq = 1.0, 1.0, 1.0
    else:  # one rate point per batch element (core/data/__init__.py:75)
        fr2 = frames(seed + 100, n_p + 1, h, w)
        xs = [torch.from_numpy(np.stack([fr[t], fr2[t]])) for t in range(n_p + 1)]
        iq = i.q_scale[:batch].detach()
        mvq = d.mv_y_q_scale[:batch].detach()
        yq = d.y_q_scale[:batch].detach()
    with torch.no_grad():
        itap.reset()
        ri = i(xs[0], iq)
        for k in ("mse", "bpp", "bpp_y", "bpp_z"):
            fx[f"i_{k}"] = ri[k].numpy()
        fx["i_bit"] = np.float64(ri["bit"])
        fx["i_xhat_stats"] = stats(ri["x_hat"])
        fx["i_xhat_crop"] = crop(ri["x_hat"])
        if batch == 1:
            for k, v in planes(i, itap, ["hyper_enc"]).items():
                fx["i_" + k.replace("hyper_enc", "z")] = v
        dpb = {"ref_frame": ri["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        for t in range(1, n_p + 1):
            dtap.reset()
            r = d.forward_one_frame(xs[t], dpb, mvq, yq)
            dpb = r["dpb"]
            p = f"p{t}_"
            for k in ("bpp_mv_y", "bpp_mv_z", "bpp_y", "bpp_z", "bpp", "me_mse", "mse"):
                fx[p + k] = r[k].numpy()
            for k in ("bit", "bit_y", "bit_z", "bit_mv_y", "bit_mv_z"):
                fx[p + k] = np.float64(r[k].item())
            for k, v in dpb.items():
                fx[p + k + "_stats"] = stats(v)
                fx[p + k + "_crop"] = crop(v)
            fx[p + "est_mv_stats"] = stats(dtap.out["optic_flow"])
            fx[p + "est_mv_crop"] = crop(dtap.out["optic_flow"])
            fx[p + "mv_hat_stats"] = stats(dtap.out["mv_decoder"])
            for ci, c in enumerate(dtap.out["context_fusion_net"], 1):
                fx[p + f"c{ci}_stats"] = stats(c)
            fx[p + "ctxdec_stats"] = stats(dtap.out["contextual_decoder"])
            if batch == 1:
                pl = planes(d, dtap, ["mv_hyper_prior_encoder", "contextual_hyper_prior_encoder"])
                for k, v in pl.items():
                    k = k.replace("mv_hyper_prior_encoder", "mv_z").replace("contextual_hyper_prior_encoder", "z")
                    fx[p + k] = v
            if h <= 64 and batch == 1:
                fx[p + "recon_full"] = dpb["ref_frame"].numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(name, {k: (v if np.ndim(v) == 0 else v.shape) for k, v in fx.items() if "bpp" in k and k.endswith("bpp")})


def table_case(d, i):
    fx = {}
    dummy = object()
    for tag, net in (("dmc", d), ("intra", i)):
        net.gaussian_encoder.update(force=True, entropy_coder=dummy)
        c, l, o = net.gaussian_encoder.cdf_helper.get_cdf_info()
        fx[f"{tag}_scale_cdf"], fx[f"{tag}_scale_len"], fx[f"{tag}_scale_off"] = c, l, o
        net.bit_estimator_z.update(force=True, entropy_coder=dummy)
        c, l, o = net.bit_estimator_z.cdf_helper.get_cdf_info()
        fx[f"{tag}_z_cdf"], fx[f"{tag}_z_len"], fx[f"{tag}_z_off"] = c, l, o
    d.bit_estimator_z_mv.update(force=True, entropy_coder=dummy)
    c, l, o = d.bit_estimator_z_mv.cdf_helper.get_cdf_info()
    fx["dmc_zmv_cdf"], fx["dmc_zmv_len"], fx["dmc_zmv_off"] = c, l, o
    # build_indexes on a sweep of scales incl. bin edges, zeros and negatives
    s = torch.cat([torch.tensor([-1.0, 0.0, 1e-6, 1e-5, 0.0099, 0.01, 0.0101, 0.11, 63.9, 64.0, 65.0, 1e4]), torch.exp(torch.linspace(-6, 5, 500))])
    fx["idx_sweep_in"] = s.numpy()
    fx["idx_sweep_laplace"] = d.gaussian_encoder.build_indexes(s.clone()).numpy()
    fx["idx_sweep_gauss"] = i.gaussian_encoder.build_indexes(s.clone()).numpy()
    # pmf_to_quantized_cdf vectors straight from the reference's ops.cpp (oracle/_ref)
    import MLCodec_CXX

    g = np.random.default_rng(5)
    vecs = [[0.1, 0.2, 0.4, 0.2, 0.05, 0.05], [1e-9] * 40 + [0.5, 0.5], [0.25, 0.25, 0.25, 0.25]]
    for n in (7, 33, 103):
        p = g.random(n).astype(np.float32) ** 6
        vecs.append((p / p.sum()).tolist())
    for k, v in enumerate(vecs):
        fx[f"pmf_{k}"] = np.array(v, np.float32)
        fx[f"qcdf_{k}"] = np.array(MLCodec_CXX.pmf_to_quantized_cdf(np.array(v, np.float32).tolist(), 16), np.int64)
    fx["n_pmf"] = np.int64(len(vecs))
    np.savez_compressed(os.path.join(OUT, "tables.npz"), **fx)
    print("tables", fx["dmc_scale_cdf"].shape, fx["dmc_z_cdf"].shape, fx["intra_scale_cdf"].shape, fx["intra_z_cdf"].shape)


def warp_case():
    from DCVC_HEM.src.models.video_net import flow_warp, bilinearupsacling, bilineardownsacling

    fx = {}
    g = np.random.default_rng(3)
    for k, (c, h, w, amp) in enumerate(((3, 17, 23, 2.0), (8, 32, 48, 40.0), (64, 16, 16, 0.7), (2, 9, 1920, 3.0))):
        im = torch.from_numpy(g.standard_normal((1, c, h, w)).astype(np.float32))
        fl = torch.from_numpy((amp * g.standard_normal((1, 2, h, w))).astype(np.float32))
        fx[f"warp{k}_im"], fx[f"warp{k}_flow"] = im.numpy(), fl.numpy()
        fx[f"warp{k}_out"] = flow_warp(im, fl).numpy()
    x = torch.from_numpy(g.standard_normal((2, 2, 6, 10)).astype(np.float32))
    fx["resamp_in"] = x.numpy()
    fx["resamp_up"] = bilinearupsacling(x).numpy()
    fx["resamp_down"] = bilineardownsacling(x).numpy()
    fx["n_warp"] = np.int64(4)
    np.savez_compressed(os.path.join(OUT, "warp.npz"), **fx)
    print("warp done")


def stream_case():
    from DCVC_HEM.src.utils import stream_helper as sh

    fx = {}
    sizes = [(1080, 1920), (256, 256), (64, 64), (65, 63), (720, 1280), (1, 1), (2160, 3840)]
    fx["sizes"] = np.array(sizes)
    fx["padding"] = np.array([sh.get_padding_size(h, w) for h, w in sizes])
    fx["down16"] = np.array([sh.get_downsampled_shape(h, w, 16) for h, w in sizes])
    fx["down64"] = np.array([sh.get_downsampled_shape(h, w, 64) for h, w in sizes])
    qs = [0.0, 0.004, 0.01, 0.5, 0.995, 1.0, 1.234567, 1.4, 654.999, 655.0, 700.0]
    fx["q_in"] = np.array(qs)
    rq = [sh.get_rounded_q(q) for q in qs]
    fx["q_scale"] = np.array([a for a, _ in rq])
    fx["q_index"] = np.array([b for _, b in rq])
    payload = bytes(range(37))
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "p.bin")
        sh.encode_p(payload, 123, 45678, p)
        fx["bin_p"] = np.frombuffer(open(p, "rb").read(), np.uint8)
        assert sh.decode_p(p) == (123, 45678, payload)
        sh.encode_i(1080, 1920, 150, payload, p)
        fx["bin_i"] = np.frombuffer(open(p, "rb").read(), np.uint8)
    fx["payload"] = np.frombuffer(payload, np.uint8)
    np.savez_compressed(os.path.join(OUT, "stream.npz"), **fx)
    print("stream done")


def second_weight_set():
    """seq_128x192_w5.npz: I + 2 P pictures with a different weight set (seed 5, gain 1.2: lower rate)."""
    d, i = build_nets(seed=5, gain=1.2)
    dtap = Tap(d, ["optic_flow", "mv_decoder", "context_fusion_net", "contextual_decoder", "mv_hyper_prior_encoder", "contextual_hyper_prior_encoder"])
    itap = Tap(i, ["hyper_enc"])
    sequence_case(d, i, dtap, itap, "seq_128x192_w5", 128, 192, 2, seed=6)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "w5":
        torch.set_num_threads(8)
        return second_weight_set()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    d, i = build_nets()
    dtap = Tap(d, ["optic_flow", "mv_decoder", "context_fusion_net", "contextual_decoder", "mv_hyper_prior_encoder", "contextual_hyper_prior_encoder"])
    itap = Tap(i, ["hyper_enc"])
    sequence_case(d, i, dtap, itap, "seq_64", 64, 64, 2, seed=0)
    sequence_case(d, i, dtap, itap, "seq_128", 128, 128, 2, seed=1)
    sequence_case(d, i, dtap, itap, "seq_256", 256, 256, 2, seed=2)
    sequence_case(d, i, dtap, itap, "seq_192x320", 192, 320, 1, seed=3)
    sequence_case(d, i, dtap, itap, "seq_64_b2", 64, 64, 2, seed=4, batch=2)
    table_case(d, i)
    warp_case()
    stream_case()


if __name__ == "__main__":
    main()
