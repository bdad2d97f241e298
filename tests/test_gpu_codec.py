"""End-to-end parity of the operator API (DMC / IntraNoAR on the HIP kernels) on a real MI355X.

* against fixtures produced by the REFERENCE itself (tests/golden/seq_*.npz) and against the
  CPU oracle on the same seeded inputs: bpp / mse within 1e-4 relative (north_star's
  tolerance for the float transforms), symbol planes equal up to the rare float->int hinge;
* bitstream: compress -> decompress reproduces the encoder's DPB bit for bit, the payload
  decodes with the oracle's independent decoders into the planes the encoder produced;
* at BASELINE's full size (1088x1920) through size-independent properties: determinism,
  encode -> decode identity, payload size vs the entropy estimate.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dcvc_ref as R
from oracle import rans_py
from tests.util import golden, oracle_weights, stats
from vcm_ts_amd.synthetic import frames

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module", params=["fp32", "fp16x3"])
def nets(request):
    """Both arithmetic modes of the convolution kernel (include/dcvc_hip.h DCVC_PREC_*) must
    meet the same tolerances: fp32 = exact fp32 MFMA, fp16x3 = split-fp16 MFMA."""
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.intra import IntraNoAR

    dev = torch.device("cuda:0")
    d, i = DMC(precision=request.param).to(dev).eval(), IntraNoAR(precision=request.param).to(dev).eval()
    d.update()
    i.update()
    return d, i


def _plane_report(got_planes, fx, prefix, report):
    """Integer planes of one picture against the planes the REFERENCE produced for it (make_golden.py
    `planes`): exact mismatch count per plane.  A symbol is round(float), an index a step function of
    a float scale, so a plane can differ only where this implementation's fp32 sums land on the other
    side of a rounding tie / bin edge than the reference's (different summation order); with
    bit-identical floats they are equal (test_build_indexes_bit_exact_against_reference_planes,
    test_dual_prior_matches_oracle)."""
    for tag, t in got_planes.items():
        want = fx[prefix + tag]
        got = t.cpu().numpy().reshape(want.shape).astype(np.int64)
        diff = got != want.astype(np.int64)
        report[prefix + tag] = (int(diff.sum()), diff.size, int(np.abs(got - want)[diff].max()) if diff.any() else 0)


def _close(got, want, tol=TOL, msg=""):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    np.testing.assert_allclose(got, want, rtol=tol, err_msg=msg)


def _check_sequence(d, i, fx, xs, n_p, name, plane_bound, deep_tol=TOL, deep_total_tol=TOL):
    """I picture + n_p P pictures through the estimate path (scalars, DPB statistics) and through
    compress (integer planes) against a fixture produced by the REFERENCE (tests/golden/make_golden*.py)."""
    h, w = xs[0].shape[-2:]
    report = {}
    vi = i.compress(xs[0], 1.0)["_views"]
    _plane_report({"sym_z": vi["sym_z"], "sym_y0": vi["r"]["sym"][0], "sym_y1": vi["r"]["sym"][1],
                   "idx_y0": vi["r"]["idx"][0], "idx_y1": vi["r"]["idx"][1]}, fx, "i_", report)
    ri = i(xs[0], 1.0)
    for k in ("mse", "bpp", "bpp_y", "bpp_z"):
        _close(ri[k], fx[f"i_{k}"], msg="i_" + k)
    assert abs(ri["bit"] - float(fx["i_bit"])) <= TOL * float(fx["i_bit"])
    if "i_xhat_mid" in fx:
        np.testing.assert_allclose(ri["x_hat"][..., 512:576, 960:1024].cpu().numpy(), fx["i_xhat_mid"], atol=2e-4)
    dpb = {"ref_frame": ri["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for t in range(1, n_p + 1):
        v = d.compress(xs[t], dpb, 1.0, 1.0)["_views"]  # same analysis as the estimate path, planes kept
        _plane_report({"sym_mv_z": v["sym_mv_z"], "sym_mv_y0": v["r_mv"]["sym"][0], "sym_mv_y1": v["r_mv"]["sym"][1],
                       "idx_mv_y0": v["r_mv"]["idx"][0], "idx_mv_y1": v["r_mv"]["idx"][1], "sym_z": v["sym_z"],
                       "sym_y0": v["r_y"]["sym"][0], "sym_y1": v["r_y"]["sym"][1], "idx_y0": v["r_y"]["idx"][0],
                       "idx_y1": v["r_y"]["idx"][1]}, fx, f"p{t}_", report)
        r = d.forward_one_frame(xs[t], dpb, 1.0, 1.0)
        dpb = r["dpb"]
        p = f"p{t}_"
        # the first P picture sees only the I picture's reconstruction: north_star's 1e-4.  Deeper pictures
        # inherit every rounding tie that fell the other way before them through the DPB (ref_feature, ref_y,
        # ref_mv_y): `deep_tol` (1e-4 for the small fixtures; stated per test where it is wider)
        tol = TOL if t == 1 else deep_tol
        devs = {}
        for k in ("bpp_mv_y", "bpp_mv_z", "bpp_y", "bpp_z", "bpp", "me_mse", "mse", "bit", "bit_y", "bit_z", "bit_mv_y", "bit_mv_z"):
            got = float(r[k].reshape(-1)[0]) if torch.is_tensor(r[k]) else float(r[k])
            want = float(np.asarray(fx[p + k]).reshape(-1)[0])
            devs[k] = abs(got - want) / abs(want)
        psnr_got = 10 * np.log10(1.0 / r["mse"].item())
        psnr_ref = 10 * np.log10(1.0 / float(fx[p + "mse"][0]))
        devs["psnr"] = abs(psnr_got - psnr_ref) / max(abs(psnr_ref), 1.0)
        print(f"\n[{name} {d.engine().precision}] p{t} relative deviation from the reference: " +
              ", ".join(f"{k} {v:.1e}" for k, v in devs.items()))
        for k, v in devs.items():
            # the totals north_star names (bpp, distortion, PSNR) hold 1e-4 on the first P picture everywhere and at
            # every depth on the small fixtures (`deep_total_tol` says where it is wider); `tol` is for the parts
            total_tol = TOL if t == 1 else deep_total_tol
            assert v <= (total_tol if k in ("bpp", "bit", "mse", "psnr") else tol), (p + k, v, tol)
        for k, v in dpb.items():
            # mean / std tightly; the abs-max is a single element and moves when one symbol rounds
            # the other way (summation-order noise of ~1e-7 is enough)
            # (the mean of a zero-centred tensor is judged on the scale of its spread, not of itself)
            want_mean, want_std = fx[p + k + "_stats"][:2]
            sf = tol / TOL
            assert abs(stats(v)[0] - want_mean) <= 2e-4 * sf * max(abs(want_mean), want_std), (p + k, stats(v)[0], want_mean)
            np.testing.assert_allclose(stats(v)[1], want_std, rtol=2e-4 * sf, err_msg=p + k)
            np.testing.assert_allclose(stats(v)[2], fx[p + k + "_stats"][2], rtol=5e-3 * sf, err_msg=p + k)
            got_c, want_c = v[..., :8, :8].cpu().numpy(), fx[p + k + "_crop"]
            bad = np.abs(got_c - want_c) > sf * (2e-4 + 2e-3 * np.abs(want_c))
            # element-wise agreement except in the neighbourhood of a symbol that rounded the other
            # way (the rate / distortion scalars above are the hard 1e-4 criterion)
            assert bad.mean() < 0.05, (p + k, bad.mean())
        assert set(r) >= {"bpp_mv_y", "bpp_mv_z", "bpp_y", "bpp_z", "bpp", "me_mse", "mse", "dpb", "bit", "bit_y",
                          "bit_z", "bit_mv_y", "bit_mv_z"}
        assert r["dpb"]["ref_feature"].shape == (1, 64, h, w) and r["dpb"]["ref_y"].shape == (1, 96, h // 16, w // 16)
    # plane-by-plane verdict against the reference's own integer planes
    print(f"\n[{name} {d.engine().precision}] mismatching elements per plane (count / size, max |delta|):")
    for k, (bad, n, mx) in report.items():
        print(f"  {k:14s} {bad:6d} / {n:8d}  max {mx}")
    # Planes are integers derived from floats (round / bin edge).  This implementation sums a
    # convolution in a different order than the reference's CPU kernels (~1e-7 relative), so a value
    # within that distance of a tie lands on the other side: a "hinge" flip, which moves the integer by
    # exactly one step.  Bound for a plane's own hinge flips: `plane_bound` of its elements.  A flipped
    # symbol of the first checkerboard half then changes y_hat_0 at that position, and the spatial
    # prior (three 3x3 convolutions, common_model.py:139-153) spreads that over its 7x7 x C/2
    # receptive field: the second half's planes are allowed that many further differences per flip.
    # The motion planes feed everything after them, so the residual planes are only judged when the
    # motion symbols are identical (they are in every committed case).  With bit-identical float
    # inputs every plane is identical (test_build_indexes_bit_exact_against_reference_planes,
    # test_dual_prior_matches_oracle); observed counts: printed above, recorded in DESIGN.md section 2.
    def judge(pic, grp, chans):
        get = lambda tag: report.get(f"{pic}_{tag}", (0, 1, 0))
        z_bad, z_n, z_mx = get(f"sym_{grp}z")
        assert z_bad <= max(1, int(z_n * plane_bound)) and z_mx <= 1, (pic, grp, "z", z_bad)
        if z_bad:
            return  # the hyper latent feeds every scale and mean of the group
        first = 0
        for half in (0, 1):
            for kind in ("sym", "idx"):
                bad, n, mx = get(f"{kind}_{grp}y{half}")
                hinge = max(2, int(n * plane_bound))
                spill = 49 * chans * first if half == 1 else 0
                assert bad <= hinge + spill, (pic, f"{kind}_{grp}y{half}", bad, hinge, spill)
                if kind == "sym":
                    assert mx <= 1, (pic, f"{kind}_{grp}y{half}", mx)
                if half == 0:
                    first += bad

    judge("i", "", 96)                      # the I picture: nothing upstream
    if "p1_sym_mv_y0" in report:
        judge("p1", "mv_", 32)              # motion planes of the first P picture: only the I reconstruction upstream
    for k, (bad, n, mx) in report.items():  # everything else sits behind earlier flips (motion, DPB): sanity bounds
        if "_sym_" in k:
            assert mx <= 1 and bad <= max(2, int(0.03 * n)), (k, bad, n, mx)


@pytest.mark.parametrize("name,h,w,n_p,seed", [("seq_64", 64, 64, 2, 0), ("seq_128", 128, 128, 2, 1),
                                               ("seq_192x320", 192, 320, 1, 3), ("seq_256", 256, 256, 2, 2)])
def test_estimate_path_matches_reference_fixtures(nets, name, h, w, n_p, seed):
    d, i = nets
    fr = frames(seed, n_p + 1, h, w)
    xs = [torch.from_numpy(fr[t : t + 1]).cuda() for t in range(n_p + 1)]
    _check_sequence(d, i, golden(name), xs, n_p, name, plane_bound=2e-3)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_second_weight_set_matches_reference_fixture(precision):
    """tests/golden/seq_128x192_w5.npz: the reference run with ANOTHER weight set (seed 5, gain 1.2; P pictures at
    0.3-0.4 bpp instead of 4-7): scalars at 1e-4, planes counted, in both arithmetic modes."""
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.intra import IntraNoAR

    dev = torch.device("cuda:0")
    d, i = DMC(precision=precision).to(dev).eval(), IntraNoAR(precision=precision).to(dev).eval()
    d.load_state_dict(oracle_weights("dmc", 5, 1.2))
    i.load_state_dict(oracle_weights("intra", 5, 1.2))
    d.update(force=True)
    i.update(force=True)
    fr = frames(6, 3, 128, 192)
    xs = [torch.from_numpy(fr[t : t + 1]).cuda() for t in range(3)]
    _check_sequence(d, i, golden("seq_128x192_w5"), xs, 2, "seq_128x192_w5", plane_bound=2e-3)


def test_bench_size_matches_reference_fixture(nets):
    """BASELINE configs[1]'s picture size, 1920x1080 zero-padded to 1088x1920 (video_coder.py:111-117),
    I + 2 P pictures against tests/golden/seq_1088x1920.npz, which make_golden_1080p.py produced by
    running the reference itself on the CPU: every rate / distortion scalar and PSNR of the I picture and
    of the first P picture within 1e-4 in BOTH arithmetic modes (second P picture: 5e-4, see below),
    integer planes counted against the reference's planes."""
    from vcm_ts_amd.pipeline import pad_frame

    d, i = nets
    fx = golden("seq_1088x1920")
    fr = frames(int(fx["seed"]), 3, int(fx["height"]), int(fx["width"]))
    xs = [pad_frame(torch.from_numpy(fr[t : t + 1])).cuda() for t in range(3)]
    assert xs[0].shape == (1, 3, 1088, 1920)
    # second P picture: at this size and rate (bpp ~6 with random-init weights, 1.4 M symbols per picture) the
    # ties of the first two pictures move single rate COMPONENTS by up to ~3e-4 (bpp_mv_y 2.4e-4 in the exact-fp32
    # mode).  Total bpp, mse and PSNR of the second P picture: within north_star's 1e-4 in BOTH modes (3e-6 exact,
    # 9.1e-5 fast).  Round 3 asserted the fast mode at 2e-4 because two of its builds gave 0.90e-4 and 1.18e-4 -- their
    # only arithmetic difference was the order in which a workgroup's waves added their SELayer partial sums; since
    # round 4 that order belongs to the tile (test_k32_wave_count_changes_no_bit), the figure no longer depends on a
    # tuning knob, and the bound is back at 1e-4.  DESIGN.md section 2 has the whole GOP-8 curve
    _check_sequence(d, i, fx, xs, 2, "seq_1088x1920", plane_bound=5e-3, deep_tol=5e-4, deep_total_tol=TOL)
    d.engine().release()
    i.engine().release()
    torch.cuda.empty_cache()


def test_bench_size_gop8_curve_against_reference(nets):
    """I + 7 P pictures at 1088x1920 against the reference's own run (tests/golden/seq_1088x1920.npz, extended to 7 P
    pictures in round 3): how the deviation grows with depth in the GOP.  Per picture: relative deviation of bpp, mse
    and PSNR from the reference, and the fraction of integer symbols that differ from the reference's planes (always by
    +-1: rounding ties that fall the other way and cascade through the checkerboard and the DPB).  The curve is printed
    and kept under profiles/ (r03_gop8_vs_reference.txt); what is asserted is stated at the end."""
    from vcm_ts_amd.pipeline import pad_frame

    d, i = nets
    fx = golden("seq_1088x1920")
    n_p = max(int(k[1]) for k in fx.files if k.startswith("p") and k[1].isdigit() and k.endswith("_bpp"))
    assert n_p >= 7
    fr = frames(int(fx["seed"]), n_p + 1, int(fx["height"]), int(fx["width"]))
    xs = [pad_frame(torch.from_numpy(fr[t : t + 1])).cuda() for t in range(n_p + 1)]
    ri = i(xs[0], 1.0)
    dpb = {"ref_frame": ri["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    lines = [f"# {d.engine().precision}: picture, rel. deviation of bpp / mse / PSNR from the reference, differing symbols: motion, residual (fraction)"]
    curve = []
    for t in range(1, n_p + 1):
        v = d.compress(xs[t], dpb, 1.0, 1.0)["_views"]
        report = {}
        _plane_report({"sym_mv_y0": v["r_mv"]["sym"][0], "sym_mv_y1": v["r_mv"]["sym"][1], "sym_y0": v["r_y"]["sym"][0],
                       "sym_y1": v["r_y"]["sym"][1]}, fx, f"p{t}_", report)
        r = d.forward_one_frame(xs[t], dpb, 1.0, 1.0)
        dpb = r["dpb"]
        dev = {}
        for k in ("bpp", "bit", "mse"):
            got = float(r[k].reshape(-1)[0]) if torch.is_tensor(r[k]) else float(r[k])
            want = float(np.asarray(fx[f"p{t}_{k}"]).reshape(-1)[0])
            dev[k] = abs(got - want) / abs(want)
        psnr_ref = 10 * np.log10(1.0 / float(fx[f"p{t}_mse"][0]))
        dev["psnr"] = abs(10 * np.log10(1.0 / r["mse"].item()) - psnr_ref) / max(abs(psnr_ref), 1.0)
        mv = sum(report[f"p{t}_sym_mv_y{h}"][0] for h in (0, 1)) / sum(report[f"p{t}_sym_mv_y{h}"][1] for h in (0, 1))
        yy = sum(report[f"p{t}_sym_y{h}"][0] for h in (0, 1)) / sum(report[f"p{t}_sym_y{h}"][1] for h in (0, 1))
        mx = max(report[k][2] for k in report)
        lines.append(f"P{t}: bpp {dev['bpp']:.1e}  mse {dev['mse']:.1e}  PSNR {dev['psnr']:.1e}   symbols: motion {mv:.2e}  residual {yy:.2e}  max |delta| {mx}")
        curve.append(dev)
    print("\n" + "\n".join(lines))
    out = os.environ.get("DCVC_CURVE_OUT")
    if out:
        with open(out, "a") as f:
            f.write("\n".join(lines) + "\n")
    # What holds, and is asserted: the first and the second P picture within north_star's 1e-4 on every total in both
    # modes (test_bench_size_matches_reference_fixture says why the fast mode's second picture is back at 1e-4).  From the third picture on the rounding ties of the earlier pictures
    # have cascaded through the DPB and single pictures move by a few 1e-4 -- in the EXACT-fp32 mode as much as in the
    # split-fp16 mode (summation order against the CPU, not operand precision: DESIGN.md section 2) -- so deeper
    # pictures get a sanity bound, and the quantity a GOP-level comparison sees, the mean over the pictures, is bounded.
    for t, dev in enumerate(curve, 1):
        assert max(dev.values()) <= (TOL if t <= 2 else 1e-3), (t, dev)
    # what a GOP-level comparison sees: the mean over the pictures (2.6e-4 fast / 1.7e-4 exact measured in round 4)
    assert float(np.mean([max(dev.values()) for dev in curve])) <= 4e-4, curve
    d.engine().release()
    i.engine().release()
    torch.cuda.empty_cache()


class _ForcedSymbols:
    """Teacher forcing: the decoder networks of `codec` are fed the REFERENCE's integer planes of picture `prefix`
    (tests/golden/seq_1088x1920*.npz hold every plane of every picture) in bitstream order instead of what a range
    decoder would return.  DPB_t is a function of (DPB_{t-1}, symbols of picture t) through float networks only -- no
    rounding in between -- so the forced decoder's DPB tracks the reference's DPB to float accuracy at every depth,
    without the cascade of rounding ties a free-running encoder accumulates."""

    def __init__(self, codec, fx, prefix, scale_tags):
        self.codec, self.fx, self.prefix, self.tags = codec, fx, prefix, list(scale_tags)

    def _plane(self, key):
        return torch.from_numpy(self.fx[self.prefix + key].astype(np.int32).reshape(-1)).to(self.codec.device)

    def __enter__(self):
        c = self.codec
        c._decode_factorized = lambda name, N, C_, H, W: self._plane("sym_mv_z" if name.endswith("_mv") else "sym_z")
        c._decode_scale = lambda idx: self._plane("sym_" + self.tags.pop(0))
        c._clamp_decoded = False  # the fixture's DPB recursion is forward_one_frame's: unclamped (video_model.py:535)
        return self

    def __exit__(self, *exc):
        for k in ("_decode_factorized", "_decode_scale", "_clamp_decoded"):
            self.codec.__dict__.pop(k, None)
        assert not self.tags or exc[0] is not None, self.tags


def _weights_for(fixture, d, i):
    if fixture.endswith("_w5"):
        d.load_state_dict(oracle_weights("dmc", 5, 1.2))
        i.load_state_dict(oracle_weights("intra", 5, 1.2))
        d.update(force=True)
        i.update(force=True)


def _codecs(precision):
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.intra import IntraNoAR

    dev = torch.device("cuda:0")
    d, i = DMC(precision=precision).to(dev).eval(), IntraNoAR(precision=precision).to(dev).eval()
    d.update()
    i.update()
    return d, i


_SCALARS = ("bpp_mv_y", "bpp_mv_z", "bpp_y", "bpp_z", "bpp", "me_mse", "mse", "bit", "bit_y", "bit_z", "bit_mv_y", "bit_mv_z")


def _scalar_devs(r, fx, p):
    devs = {}
    for k in _SCALARS:
        got = float(r[k].reshape(-1)[0]) if torch.is_tensor(r[k]) else float(r[k])
        want = float(np.asarray(fx[p + k]).reshape(-1)[0])
        devs[k] = abs(got - want) / abs(want)
    psnr_ref = 10 * np.log10(1.0 / float(fx[p + "mse"][0]))
    devs["psnr"] = abs(10 * np.log10(1.0 / r["mse"].item()) - psnr_ref) / max(abs(psnr_ref), 1.0)
    return devs


@pytest.mark.parametrize("fixture", ["seq_1088x1920", "seq_1088x1920_w5"])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_bench_size_teacher_forced_every_depth(precision, fixture):
    """Per-picture parity at BASELINE configs[1]'s size at EVERY depth of the reference's I + 7 P run, free of the GOP
    recursion.  For t = 1..7 the encoder is given the reference's own DPB_{t-1} -- rebuilt on the
    GPU by the decoder networks from the reference's symbol planes (_ForcedSymbols), and checked against the
    reference's DPB statistics and crops -- and every rate / distortion scalar of picture t is compared with the
    reference's p{t}_* at north_star's 1e-4, in BOTH arithmetic modes; integer planes are counted against the
    reference's with the hinge bound of _check_sequence (each picture now only sees its own ties)."""
    from vcm_ts_amd.pipeline import pad_frame

    d, i = _codecs(precision)
    _weights_for(fixture, d, i)
    fx = golden(fixture)
    n_p = max(int(k[1]) for k in fx.files if k.startswith("p") and k[1].isdigit() and k.endswith("_bpp"))
    fr = frames(int(fx["seed"]), n_p + 1, int(fx["height"]), int(fx["width"]))
    xs = [pad_frame(torch.from_numpy(fr[t : t + 1])).cuda() for t in range(n_p + 1)]
    h, w = int(fx["height"]), int(fx["width"])
    dummy = bytes(16)
    with _ForcedSymbols(i, fx, "i_", ["y0", "y1"]):
        x_hat = i.decompress(dummy, h, w, 1.0, coder="host")["x_hat"].clone()
    np.testing.assert_allclose(x_hat[..., 512:576, 960:1024].cpu().numpy(), fx["i_xhat_mid"], atol=1e-5)
    np.testing.assert_allclose(stats(x_hat)[:2], fx["i_xhat_stats"][:2], rtol=1e-5)
    dpb = {"ref_frame": x_hat, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    lines = [f"# {fixture} {precision}, teacher-forced: picture, worst relative deviation of a total (bpp/bit/mse/PSNR), of a component, differing symbols motion / residual"]
    worst = 0.0
    for t in range(1, n_p + 1):
        p = f"p{t}_"
        v = d.compress(xs[t], dpb, 1.0, 1.0)["_views"]
        report = {}
        _plane_report({"sym_mv_z": v["sym_mv_z"], "sym_mv_y0": v["r_mv"]["sym"][0], "sym_mv_y1": v["r_mv"]["sym"][1],
                       "idx_mv_y0": v["r_mv"]["idx"][0], "idx_mv_y1": v["r_mv"]["idx"][1], "sym_z": v["sym_z"],
                       "sym_y0": v["r_y"]["sym"][0], "sym_y1": v["r_y"]["sym"][1], "idx_y0": v["r_y"]["idx"][0],
                       "idx_y1": v["r_y"]["idx"][1]}, fx, p, report)
        r = d.forward_one_frame(xs[t], dpb, 1.0, 1.0)
        devs = _scalar_devs(r, fx, p)
        tot = max(devs[k] for k in ("bpp", "bit", "mse", "psnr"))
        comp = max(devs.values())
        mv = sum(report[p + f"sym_mv_y{k}"][0] for k in (0, 1)) / sum(report[p + f"sym_mv_y{k}"][1] for k in (0, 1))
        yy = sum(report[p + f"sym_y{k}"][0] for k in (0, 1)) / sum(report[p + f"sym_y{k}"][1] for k in (0, 1))
        lines.append(f"P{t}: totals {tot:.1e}  components {comp:.1e}   symbols: motion {mv:.2e}  residual {yy:.2e}  "
                     f"max |delta| {max(x[2] for x in report.values())}")
        # north_star's tolerance, at every depth, on the totals AND on every component, in both modes
        for k, dv in devs.items():
            assert dv <= TOL, (fixture, precision, p + k, dv)
        worst = max(worst, comp)
        # planes: every symbol delta is one step; motion planes within the hinge bound (they only see DPB_{t-1})
        for k, (bad, n, mx) in report.items():
            if "_sym_" in k:
                assert mx <= 1, (k, mx)
            if "_mv_" in k:
                assert bad <= max(2, int(5e-3 * n)) + (49 * 32 * report[p + "sym_mv_y0"][0] if k.endswith("1") else 0), (k, bad, n)
            assert bad <= max(2, int(0.03 * n)), (k, bad, n)
        # the reference's DPB_t through the decoder networks, from the reference's symbols of picture t
        with _ForcedSymbols(d, fx, p, ["mv_y0", "mv_y1", "y0", "y1"]):
            nxt = d.decompress(dpb, dummy, h, w, 1.0, 1.0, coder="host")["dpb"]
        nxt = {k: t_.clone() for k, t_ in nxt.items()}
        for k, t_ in nxt.items():
            want_mean, want_std, want_max = fx[p + k + "_stats"][:3]
            assert abs(stats(t_)[0] - want_mean) <= 2e-5 * max(abs(want_mean), want_std), (p + k, stats(t_)[0], want_mean)
            np.testing.assert_allclose(stats(t_)[1], want_std, rtol=2e-5, err_msg=p + k)
            np.testing.assert_allclose(stats(t_)[2], want_max, rtol=1e-4, err_msg=p + k)
            np.testing.assert_allclose(t_[..., :8, :8].cpu().numpy(), fx[p + k + "_crop"], rtol=2e-4, atol=2e-5, err_msg=p + k)
        # (float noise of ~50 layers between two summation orders: 6e-5 absolute on values of +-1.3 measured)
        np.testing.assert_allclose(nxt["ref_frame"][..., 512:576, 960:1024].cpu().numpy(), fx[p + "recon_mid"], rtol=2e-4, atol=2e-4)
        dpb = nxt
    print("\n" + "\n".join(lines))
    out = os.environ.get("DCVC_CURVE_OUT")
    if out:
        with open(out, "a") as f:
            f.write("\n".join(lines) + "\n")
    d.engine().release()
    i.engine().release()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_bench_size_low_rate_gop8_free_running(precision):
    """tests/golden/seq_1088x1920_w5.npz (round 4): the reference's I + 7 P run at 1088x1920 with the SECOND weight set
    (seed 5, gain 1.2: P pictures at 0.3-0.4 bpp, the rate range a trained model works at).  FREE-RUNNING (each
    picture coded from this implementation's own DPB), every total within north_star's 1e-4 at every depth, both
    arithmetic modes; symbol planes counted."""
    from vcm_ts_amd.pipeline import pad_frame

    d, i = _codecs(precision)
    _weights_for("seq_1088x1920_w5", d, i)
    fx = golden("seq_1088x1920_w5")
    n_p = max(int(k[1]) for k in fx.files if k.startswith("p") and k[1].isdigit() and k.endswith("_bpp"))
    assert n_p >= 7
    fr = frames(int(fx["seed"]), n_p + 1, int(fx["height"]), int(fx["width"]))
    xs = [pad_frame(torch.from_numpy(fr[t : t + 1])).cuda() for t in range(n_p + 1)]
    ri = i(xs[0], 1.0)
    for k in ("mse", "bpp", "bpp_y", "bpp_z"):
        _close(ri[k], fx[f"i_{k}"], msg="i_" + k)
    dpb = {"ref_frame": ri["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    lines = [f"# seq_1088x1920_w5 {precision}, free-running: picture, worst relative deviation of a total, of a component, differing symbols motion / residual"]
    for t in range(1, n_p + 1):
        p = f"p{t}_"
        v = d.compress(xs[t], dpb, 1.0, 1.0)["_views"]
        report = {}
        _plane_report({"sym_mv_y0": v["r_mv"]["sym"][0], "sym_mv_y1": v["r_mv"]["sym"][1], "sym_y0": v["r_y"]["sym"][0],
                       "sym_y1": v["r_y"]["sym"][1]}, fx, p, report)
        r = d.forward_one_frame(xs[t], dpb, 1.0, 1.0)
        dpb = r["dpb"]
        devs = _scalar_devs(r, fx, p)
        tot = max(devs[k] for k in ("bpp", "bit", "mse", "psnr"))
        mv = sum(report[p + f"sym_mv_y{k}"][0] for k in (0, 1)) / sum(report[p + f"sym_mv_y{k}"][1] for k in (0, 1))
        yy = sum(report[p + f"sym_y{k}"][0] for k in (0, 1)) / sum(report[p + f"sym_y{k}"][1] for k in (0, 1))
        lines.append(f"P{t}: totals {tot:.1e}  components {max(devs.values()):.1e}   symbols: motion {mv:.2e}  residual {yy:.2e}  "
                     f"max |delta| {max(x[2] for x in report.values())}")
        for k in ("bpp", "bit", "mse", "psnr"):
            assert devs[k] <= TOL, (precision, p + k, devs[k])
        for k, dv in devs.items():
            assert dv <= 5e-4, (precision, p + k, dv)
    print("\n" + "\n".join(lines))
    out = os.environ.get("DCVC_CURVE_OUT")
    if out:
        with open(out, "a") as f:
            f.write("\n".join(lines) + "\n")
    d.engine().release()
    i.engine().release()
    torch.cuda.empty_cache()


def test_batch_of_rate_points_matches_reference(nets):
    d, i = nets
    fx = golden("seq_64_b2")
    fr, fr2 = frames(4, 3, 64, 64), frames(104, 3, 64, 64)
    xs = [torch.from_numpy(np.stack([fr[t], fr2[t]])).cuda() for t in range(3)]
    ri = i(xs[0], i.q_scale[:2])
    _close(ri["bpp"], fx["i_bpp"])
    dpb = {"ref_frame": ri["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for t in (1, 2):
        r = d.forward_one_frame(xs[t], dpb, d.mv_y_q_scale[:2], d.y_q_scale[:2])
        dpb = r["dpb"]
        for k in ("bpp", "mse", "me_mse", "bpp_y", "bpp_mv_y"):
            _close(r[k], fx[f"p{t}_{k}"], msg=k)
        assert r["mse"].shape == (2,)


def test_intermediates_and_symbols_match_oracle(nets):
    d, i = nets
    wd, wi = oracle_weights("dmc"), oracle_weights("intra")
    fr = frames(11, 3, 128, 192)
    xs = [torch.from_numpy(fr[t : t + 1]) for t in range(3)]
    with torch.no_grad():
        ro = R.intra_forward(wi, xs[0], 0.8)
        rg = i.compress(xs[0].cuda(), 0.8)
        og = rg["_views"]
        for tag, sym, sc in R.intra_symbol_planes(ro["_inter"]):
            key = {"z": og["sym_z"], "y0": og["r"]["sym"][0], "y1": og["r"]["sym"][1]}[tag]
            got = key.cpu().numpy().reshape(sym.shape)
            assert (got != sym.numpy()).mean() < 1e-4, tag
        dpb_o = {"ref_frame": ro["x_hat"].clamp(0, 1), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        dpb_g = {"ref_frame": rg["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        np.testing.assert_allclose(rg["x_hat"].cpu().numpy(), dpb_o["ref_frame"].numpy(), atol=2e-5)
        for t in (1, 2):
            po = R.dmc_forward_one_frame(wd, xs[t], dpb_o, 1.1, 0.9)
            pg = d.compress(xs[t].cuda(), dpb_g, 1.1, 0.9)
            v, o = pg["_views"], po["_inter"]
            for name in ("est_mv", "mv_hat", "c1", "c2", "c3", "y_hat", "mv_y_hat", "feature"):
                got, want = v[name].nchw().cpu(), o[name]
                assert ((got - want).abs().max() / want.abs().max()).item() < 5e-5, name
            planes = {"mv_z": v["sym_mv_z"], "mv_y0": v["r_mv"]["sym"][0], "mv_y1": v["r_mv"]["sym"][1],
                      "z": v["sym_z"], "y0": v["r_y"]["sym"][0], "y1": v["r_y"]["sym"][1]}
            for tag, sym, sc in R.dmc_symbol_planes(o):
                got = planes[tag].cpu().numpy().reshape(sym.shape)
                assert (got != sym.numpy()).mean() < 1e-4, tag
            # compress clamps the reconstruction exactly as the decoder will (video_model.py:413)
            np.testing.assert_allclose(pg["dpb"]["ref_frame"].cpu().numpy(), o["recon"].clamp(0, 1).numpy(), atol=3e-5)
            dpb_o = dict(po["dpb"], ref_frame=o["recon"].clamp(0, 1))
            dpb_g = pg["dpb"]


def _decode_with_oracle(payload, batches):
    """Decode `payload` with the oracle's pure-Python rANS into the symbol planes."""
    dec = rans_py.Decoder(payload)
    return [np.array(dec.decode(idx.tolist(), cdf, ln, off), np.int32) for idx, cdf, ln, off in batches]


def test_bitstream_round_trip_and_oracle_decode(nets, tmp_path):
    d, i = nets
    h, w = 64, 128
    fr = frames(5, 3, h, w)
    xs = [torch.from_numpy(fr[t : t + 1]).cuda() for t in range(3)]
    ci = i.compress(xs[0], 1.0)
    di = i.decompress(ci["bit_stream"], h, w, 1.0)
    assert torch.equal(di["x_hat"], ci["x_hat"])  # bit for bit
    assert float(di["x_hat"].min()) >= 0.0 and float(di["x_hat"].max()) <= 1.0
    dpb = {"ref_frame": di["x_hat"].clone(), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for t in (1, 2):
        c = d.compress(xs[t], dpb, 1.0, 1.0)
        enc_dpb = {k: v.clone() for k, v in c["dpb"].items()}
        v = c["_views"]
        planes = [v["sym_mv_z"], v["r_mv"]["sym"][0], v["r_mv"]["sym"][1], v["sym_z"], v["r_y"]["sym"][0], v["r_y"]["sym"][1]]
        planes = [p.cpu().numpy().copy() for p in planes]
        idxs = [None, v["r_mv"]["idx"][0], v["r_mv"]["idx"][1], None, v["r_y"]["idx"][0], v["r_y"]["idx"][1]]
        idxs = [None if q is None else q.cpu().numpy().copy() for q in idxs]
        dd = d.decompress(dpb, c["bit_stream"], h, w, 1.0, 1.0)
        for k in enc_dpb:
            assert torch.equal(dd["dpb"][k], enc_dpb[k]), k
        if t == 1:  # independent decode of the payload (pure-Python oracle; small picture)
            zh, zw = h // 64, w // 64
            chan = np.broadcast_to(np.arange(64, dtype=np.int32)[None, :, None, None], (1, 64, zh, zw)).reshape(-1)
            tabs = d._tables
            batches = []
            for k, name in enumerate(("bit_estimator_z_mv", "scale", "scale", "bit_estimator_z", "scale", "scale")):
                batches.append((chan if idxs[k] is None else idxs[k], *tabs[name]))
            for got, want in zip(_decode_with_oracle(c["bit_stream"], batches), planes):
                np.testing.assert_array_equal(got, want)
        dpb = {k: v.clone() for k, v in dd["dpb"].items()}
    # file round trip through encode_decode (the reference's call), header bytes included
    path = os.path.join(tmp_path, "p.bin")
    r = d.encode_decode(xs[2], dpb, path, pic_width=w, pic_height=h, mv_y_q_scale=1.234567, y_q_scale=0.5)
    raw = open(path, "rb").read()
    assert raw[:4] == bytes([0, 123, 0, 50]) and r["bit"] == len(raw) * 8
    assert set(r) == {"dpb", "bit", "encoding_time", "decoding_time"}
    r2 = d.encode_decode(xs[2], dpb, None, mv_y_q_scale=1.0, y_q_scale=1.0)
    assert set(r2) == {"dpb", "bit_y", "bit_z", "bit_mv_y", "bit_mv_z", "bit", "decoding_time"}
    ipath = os.path.join(tmp_path, "i.bin")
    ri = i.encode_decode(xs[0], 1.0, ipath, pic_width=w, pic_height=h)
    assert ri["bit"] == os.path.getsize(ipath) * 8 and torch.equal(ri["x_hat"], di["x_hat"])


def test_deferred_streams_are_guarded(nets):
    d, i = nets
    x = torch.rand(1, 3, 64, 64).cuda()
    a = i.compress(x, 1.0, defer=True)
    b = i.compress(x, 1.0, defer=True)
    with pytest.raises(RuntimeError):  # a third picture would overwrite the first one's pinned planes
        i.compress(x, 1.0, defer=True)
    assert a["pending"].finish() == b["pending"].finish() == i.compress(x, 1.0)["bit_stream"]


def test_requires_update_and_padding(nets):
    from vcm_ts_amd.dmc import DMC

    fresh = DMC().cuda().eval()
    x = torch.rand(1, 3, 64, 64).cuda()
    dpb = {"ref_frame": x, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    with pytest.raises(RuntimeError):
        fresh.compress(x, dpb, 1.0, 1.0)
    with pytest.raises(AssertionError):
        fresh.forward_one_frame(torch.rand(1, 3, 60, 64).cuda(), dpb, 1.0, 1.0)
    # the I-picture codec's training mode (round 4; tests/test_gpu_backward.py has its gradients): noisy-latent bit
    # estimate, same distortion, `bit` a float as in the reference (image_model.py:102)
    from vcm_ts_amd.intra import IntraNoAR

    inet = IntraNoAR().cuda()
    ti = inet.train()(x, 1.0)
    ei = inet.eval()(x, 1.0)
    assert isinstance(ti["bit"], float) and ti["bpp"].shape == (1,) and torch.isfinite(ti["bpp"]).all()
    torch.testing.assert_close(ti["mse"], ei["mse"], rtol=1e-5, atol=0)
    assert float((ti["bpp"] - ei["bpp"]).abs()) > 0
    # training-mode DMC forward: noisy-latent bit estimate differs from the eval estimate, same distortion
    ev = fresh(x, dpb, 1.0, 1.0)
    tr = fresh.train()(x, dpb, 1.0, 1.0)
    fresh.eval()
    assert not tr["bpp"].requires_grad and tr["bpp"].shape == (1,) and torch.isfinite(tr["bpp"]).all()
    torch.testing.assert_close(tr["mse"], ev["mse"], rtol=1e-5, atol=0)
    assert float((tr["bpp"] - ev["bpp"]).abs()) > 0


def test_full_size_properties(nets):
    """1088x1920 (padded 1080p): determinism, encode->decode identity, size vs estimate."""
    from vcm_ts_amd.pipeline import GopEncoder, pad_frame

    d, i = nets
    dev = torch.device("cuda:0")
    base = torch.from_numpy(frames(9, 1, 1080, 1920)[0]).to(dev)
    seq = [pad_frame(torch.roll(base, shifts=(t, -2 * t), dims=(1, 2))[None].contiguous()) for t in range(3)]
    assert seq[0].shape == (1, 3, 1088, 1920)
    enc = GopEncoder(i, d, gop_size=32)
    coded_a, bits_a, _ = enc.encode_gop(seq, 1.0, 1.0, 1.0)
    coded_b, bits_b, dpb_b = enc.encode_gop(seq, 1.0, 1.0, 1.0)
    assert [c[2] for c in coded_a] == [c[2] for c in coded_b] and bits_a == bits_b  # run-to-run identical bytes
    last = dpb_b["ref_frame"].clone()
    recs = enc.decode_gop(coded_a, 1080, 1920)
    assert torch.equal(recs[-1], last)  # decoder reproduces the encoder's reference picture
    # entropy estimate vs real payload on the last P picture ("usually < 0.5 %", DCVC_HEM/README.md:50)
    dpb = {"ref_frame": i.compress(seq[0], 1.0)["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    est = d.forward_one_frame(seq[1], dpb, 1.0, 1.0)
    real = len(d.compress(seq[1], dpb, 1.0, 1.0)["bit_stream"]) * 8
    assert abs(real - est["bit"].item()) / real < 0.02


def test_folder_encode_decode_round_trip(tmp_path):
    """run_dcvc-style loop on PNG folders with a size that needs padding (100x150 -> 128x192):
    decoded PNGs are identical to the encoder's own reconstructions, bits = file sizes."""
    from PIL import Image

    from vcm_ts_amd import run_codec
    from vcm_ts_amd import stream as S

    src, bins, rec_e, rec_d = (os.path.join(tmp_path, d) for d in ("src", "bins", "rec_enc", "rec_dec"))
    os.makedirs(src)
    fr = frames(21, 5, 100, 150)
    for t in range(5):
        Image.fromarray(np.clip(np.rint(fr[t].transpose(1, 2, 0) * 255), 0, 255).astype(np.uint8)).save(
            os.path.join(src, f"im{t + 1:05d}.png"))
    bits, size = run_codec.encode_folder(src, bins, rec_e, gop=3, q=(1.0, 1.1, 0.9))
    assert size == (100, 150) and len(bits) == 5
    names = sorted(os.listdir(bins))
    assert names == [f"im{t + 1:05d}.bin" for t in range(5)]
    assert bits == [os.path.getsize(os.path.join(bins, n)) * 8 for n in names]
    h, w, qi, _ = S.decode_i(os.path.join(bins, "im00001.bin"))
    assert (h, w, qi) == (100, 150, 100)
    assert S.decode_p(os.path.join(bins, "im00002.bin"))[:2] == (110, 90)
    assert S.decode_i(os.path.join(bins, "im00004.bin"))[:2] == (100, 150)  # second GOP starts with an I picture
    assert run_codec.decode_folder(bins, rec_d, 100, 150, gop=3) == 5
    for t in range(5):
        a = np.asarray(Image.open(os.path.join(rec_e, f"im{t + 1:05d}.png")))
        b = np.asarray(Image.open(os.path.join(rec_d, f"im{t + 1:05d}.png")))
        assert a.shape == (100, 150, 3) and np.array_equal(a, b)
    # the same folder through the opt-in device coder: same containers, same reconstructions
    bins2, rec_d2 = os.path.join(tmp_path, "bins_dev"), os.path.join(tmp_path, "rec_dec_dev")
    bits2, _ = run_codec.encode_folder(src, bins2, None, gop=3, q=(1.0, 1.1, 0.9), coder="device")
    assert len(bits2) == 5 and all(b2 > b1 for b1, b2 in zip(bits, bits2))  # 12 B per lane on top of the same coder
    assert S.decode_p(os.path.join(bins2, "im00002.bin"))[2][:4] == b"DGR1"
    assert run_codec.decode_folder(bins2, rec_d2, 100, 150, gop=3) == 5
    for t in range(5):
        a = np.asarray(Image.open(os.path.join(rec_d, f"im{t + 1:05d}.png")))
        b = np.asarray(Image.open(os.path.join(rec_d2, f"im{t + 1:05d}.png")))
        assert np.array_equal(a, b)


def test_folder_encode_with_two_gop_streams_writes_the_same_files(tmp_path):
    """run_codec.encode_folder(gop_streams=2): two GOPs of the folder in flight on the GPU (ConcurrentGopEncoder, one
    shared pool of PNG-decoding threads).  8 pictures at GOP 3 = GOPs 0 and 2 on stream 0 (the last
    one partial), GOP 1 on stream 1: every .bin file, the bits list and the encoder-side reconstructions are identical
    to the one-stream loop's, with reader threads and inline."""
    from PIL import Image

    from vcm_ts_amd import run_codec

    src = os.path.join(tmp_path, "src")
    os.makedirs(src)
    fr = frames(22, 8, 100, 150)
    for t in range(8):
        Image.fromarray(np.clip(np.rint(fr[t].transpose(1, 2, 0) * 255), 0, 255).astype(np.uint8)).save(
            os.path.join(src, f"im{t + 1:05d}.png"))
    out = {}
    for tag, kw in (("one", dict(gop_streams=1)), ("two", dict(gop_streams=2)), ("two_inline", dict(gop_streams=2, io_workers=0)),
                    ("three", dict(gop_streams=3, io_workers=2))):
        bins, rec = os.path.join(tmp_path, "bins_" + tag), os.path.join(tmp_path, "rec_" + tag)
        bits, size = run_codec.encode_folder(src, bins, rec, gop=3, q=(1.0, 1.1, 0.9), precision="fp16x3", **kw)
        assert size == (100, 150) and len(bits) == 8
        names = sorted(os.listdir(bins))
        assert names == [f"im{t + 1:05d}.bin" for t in range(8)]
        out[tag] = (bits, [open(os.path.join(bins, n), "rb").read() for n in names],
                    [np.asarray(Image.open(os.path.join(rec, f"im{t + 1:05d}.png"))) for t in range(8)])
    for tag in ("two", "two_inline", "three"):
        assert out[tag][0] == out["one"][0], tag
        assert out[tag][1] == out["one"][1], tag
        assert all(np.array_equal(a, b) for a, b in zip(out[tag][2], out["one"][2])), tag
    assert run_codec.decode_folder(os.path.join(tmp_path, "bins_two"), os.path.join(tmp_path, "dec"), 100, 150, gop=3,
                                   precision="fp16x3") == 8
    for t in range(8):
        assert np.array_equal(np.asarray(Image.open(os.path.join(tmp_path, "dec", f"im{t + 1:05d}.png"))), out["one"][2][t])


def test_folder_encode_fails_loudly_when_the_split_fp16_range_is_exceeded(tmp_path):
    """the fast mode clamps |activation| > 8188 on load and only FLAGS it (status word); the shipped folder
    loop now reads that flag once per GOP.  A checkpoint with an absurd bias (outputs ~1e4) must make encode_folder
    raise instead of writing .bin files and reporting success; the same checkpoint codes fine in exact-fp32 mode."""
    from PIL import Image

    from vcm_ts_amd import lib, run_codec

    src = os.path.join(tmp_path, "src")
    os.makedirs(src)
    fr = frames(23, 4, 64, 64)
    for t in range(4):
        Image.fromarray(np.clip(np.rint(fr[t].transpose(1, 2, 0) * 255), 0, 255).astype(np.uint8)).save(
            os.path.join(src, f"im{t + 1:05d}.png"))
    for precision, fails in (("fp16x3", True), ("fp32", False)):
        i_net, p_net = run_codec._nets(torch.device("cuda:0"), precision)
        with torch.no_grad():
            p_net.P("feature_extractor.conv1.bias").fill_(1.0e4)
        bins = os.path.join(tmp_path, "bins_" + precision)
        if fails:
            with pytest.raises(lib.KernelError, match="range"):
                run_codec.encode_folder(src, bins, None, gop=4, nets=(i_net, p_net))
        else:
            bits, _ = run_codec.encode_folder(src, bins, None, gop=4, nets=(i_net, p_net))
            assert len(bits) == 4


def test_batch_of_rate_points_compress(nets):
    """Variable-rate encode (BASELINE config 5 / SURVEY 8f-4): N pictures = N rate points through
    one compress call; each element's stream equals what a batch-1 call at that rate produces and
    decodes with the ordinary decoder."""
    d, i = nets
    h, w = 64, 128
    fr = frames(31, 2, h, w)
    x0 = torch.from_numpy(fr[0:1]).cuda()
    x1 = torch.from_numpy(fr[1:2]).cuda()
    qs_mv, qs_y = [1.4, 0.8], [1.2, 0.7]
    ref = i.compress(x0, 1.0)["x_hat"].clone()
    singles = []
    for k in range(2):
        dpb = {"ref_frame": ref.clone(), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        singles.append(d.compress(x1, dpb, qs_mv[k], qs_y[k])["bit_stream"])
    dpb2 = {"ref_frame": torch.cat([ref, ref]), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    r = d.compress(torch.cat([x1, x1]), dpb2, torch.tensor(qs_mv).view(2, 1, 1, 1), torch.tensor(qs_y).view(2, 1, 1, 1))
    assert len(r["bit_streams"]) == 2 and r["bit_stream"] == r["bit_streams"][0]
    assert r["bit_streams"] == singles and singles[0] != singles[1]
    rec_batch = r["dpb"]["ref_frame"].clone()
    for k in range(2):
        dpb = {"ref_frame": ref.clone(), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        dec = d.decompress(dpb, r["bit_streams"][k], h, w, qs_mv[k], qs_y[k])["dpb"]["ref_frame"]
        assert torch.equal(dec[0], rec_batch[k])


_GOP_ORACLE = {}


def _gop_oracle(fr):
    """The CPU oracle's I + 15 P run of test_gop_recursion_tracks_oracle, once for both arithmetic modes (it is
    most of that test's time): the I picture's bpp and every P picture's scalars."""
    if not _GOP_ORACLE:
        wd, wi = oracle_weights("dmc"), oracle_weights("intra")
        with torch.no_grad():
            ro = R.intra_forward(wi, torch.from_numpy(fr[0:1]), 1.0)
            _GOP_ORACLE["i_bpp"] = ro["bpp"].numpy()
            dpb_o = {"ref_frame": ro["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
            rows = []
            for t in range(1, fr.shape[0]):
                po = R.dmc_forward_one_frame(wd, torch.from_numpy(fr[t : t + 1]), dpb_o, 1.0, 1.0)
                rows.append({k: po[k].item() for k in ("bpp", "mse", "bpp_y", "bpp_mv_y")})
                dpb_o = po["dpb"]
        _GOP_ORACLE["p"] = rows
    return _GOP_ORACLE


def test_gop_recursion_tracks_oracle(nets):
    """16 pictures (I + 15 P) through the DPB recursion: every picture's bpp / mse stays within
    1e-4 of the CPU oracle run on the same inputs (no drift through ref_feature / ref_y / ref_mv_y)."""
    d, i = nets
    n, h, w = 16, 128, 192
    fr = frames(17, n, h, w)
    want = _gop_oracle(fr)
    with torch.no_grad():
        rg = i(torch.from_numpy(fr[0:1]).cuda(), 1.0)
        _close(rg["bpp"], want["i_bpp"])
        dpb_g = {"ref_frame": rg["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        worst = 0.0
        for t in range(1, n):
            pg = d.forward_one_frame(torch.from_numpy(fr[t : t + 1]).cuda(), dpb_g, 1.0, 1.0)
            for k, ref in want["p"][t - 1].items():
                rel = abs(pg[k].item() - ref) / abs(ref)
                worst = max(worst, rel)
                assert rel < TOL, (t, k, pg[k].item(), ref)
            dpb_g = pg["dpb"]
    print("worst relative deviation over the GOP:", worst)


def test_escape_coded_symbols_round_trip(nets):
    """Tiny q-scales blow the latents far outside every CDF table, so most symbols take the
    sentinel + bypass-nibble path (rans_interface.cpp:105-143 / :217-238) inside the real
    pipeline; the decoder must still reproduce the encoder's pictures bit for bit."""
    d, i = nets
    h, w = 64, 64
    fr = frames(41, 2, h, w)
    x0, x1 = torch.from_numpy(fr[0:1]).cuda(), torch.from_numpy(fr[1:2]).cuda()
    ci = i.compress(x0, 0.01)
    ysym = ci["_views"]["r"]["sym"][0].cpu().numpy()
    assert np.abs(ysym).max() > 60  # beyond the widest table (|offset| <= 50): escapes are exercised
    di = i.decompress(ci["bit_stream"], h, w, 0.01)
    assert torch.equal(di["x_hat"], ci["x_hat"])
    dpb = {"ref_frame": di["x_hat"].clone(), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    c = d.compress(x1, dpb, 0.01, 0.01)
    assert np.abs(c["_views"]["r_y"]["sym"][1].cpu().numpy()).max() > 60
    enc = {k: v.clone() for k, v in c["dpb"].items()}
    dd = d.decompress(dpb, c["bit_stream"], h, w, 0.01, 0.01)
    for k in enc:
        assert torch.equal(dd["dpb"][k], enc[k]), k


def test_graph_replay_encodes_identically(nets):
    """GopEncoder(graphs=True): P pictures replayed as captured hipGraphs give the same payload bytes
    and the same DPB as eager launches, across GOP boundaries (first-P graphs, both DPB buffer sets)."""
    from vcm_ts_amd.pipeline import GopEncoder

    d, i = nets
    dev = torch.device("cuda:0")
    fr = frames(33, 11, 128, 192)
    seq = [torch.from_numpy(fr[t : t + 1]).to(dev) for t in range(11)]
    eager, bits_e, dpb_e = GopEncoder(i, d, gop_size=4).encode_gop(seq, 1.0, 1.0, 1.0)
    ref = dpb_e["ref_frame"].clone()
    enc = GopEncoder(i, d, gop_size=4, graphs=True)
    for rep in range(2):  # second pass: pure replays
        coded, bits_g, dpb_g = enc.encode_gop(seq, 1.0, 1.0, 1.0)
        assert bits_g == bits_e and [c[2] for c in coded] == [c[2] for c in eager], rep
        assert torch.equal(dpb_g["ref_frame"], ref), rep
    assert 2 <= len(d._graphs) <= 8
    with pytest.raises(TypeError):
        d.compress(seq[1], {"ref_frame": seq[0], "ref_feature": None, "ref_y": None, "ref_mv_y": None},
                   torch.tensor(1.0), 1.0, graph=True)


def test_concurrent_gop_streams_encode_identically():
    """ConcurrentGopEncoder: two GOPs in flight on two HIP streams (own codec instances each, one host
    thread) produce exactly the payloads and reconstructions of coding them one after the other."""
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.intra import IntraNoAR
    from vcm_ts_amd.pipeline import ConcurrentGopEncoder, GopEncoder

    dev = torch.device("cuda:0")
    make = lambda: (IntraNoAR().to(dev).eval(), DMC().to(dev).eval())
    seqs = []
    for k in range(2):
        fr = frames(60 + k, 5, 128, 192)
        seqs.append([torch.from_numpy(fr[t : t + 1]).to(dev) for t in range(5)])
    cenc = ConcurrentGopEncoder(make, gop_size=5, streams=2)
    res = cenc.encode_gops(seqs, 1.0, 1.0, 1.0)
    one = GopEncoder(*make(), gop_size=5)
    for k in range(2):
        coded, bits, dpb = one.encode_gop(seqs[k], 1.0, 1.0, 1.0)
        assert [c[2] for c in coded] == [c[2] for c in res[k][0]] and bits == res[k][1], k
        assert torch.equal(dpb["ref_frame"], res[k][2]["ref_frame"]), k
    # a single sequence works too (fewer sequences than streams)
    again = cenc.encode_gops(seqs[:1], 1.0, 1.0, 1.0)
    assert [c[2] for c in again[0][0]] == [c[2] for c in res[0][0]]
    # concurrent decode (one thread + stream per GOP) == sequential decode == the encoders' reconstructions
    recs = cenc.decode_gops([r[0] for r in res], 128, 192)
    for k in range(2):
        want = one.decode_gop(res[k][0], 128, 192)
        assert len(recs[k]) == 5 and all(torch.equal(a, b) for a, b in zip(recs[k], want)), k
        assert torch.equal(recs[k][-1], res[k][2]["ref_frame"]), k


def test_concurrent_gop_streams_are_repeatable_at_a_larger_size():
    """The same property in the mode and at a size where kernels of the two streams really share the CUs (fast mode,
    512x768, GOP 4), three concurrent runs and one sequential: identical payloads every time.  (Round 2 found one
    TRAINING kernel whose result depended on what ran beside it; this guards the codec path.)"""
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.intra import IntraNoAR
    from vcm_ts_amd.pipeline import ConcurrentGopEncoder

    dev = torch.device("cuda:0")
    make = lambda: (IntraNoAR(precision="fp16x3").to(dev).eval(), DMC(precision="fp16x3").to(dev).eval())
    seqs = []
    for k in range(2):
        fr = frames(80 + k, 4, 512, 768)
        seqs.append([torch.from_numpy(fr[t : t + 1]).to(dev) for t in range(4)])
    cenc = ConcurrentGopEncoder(make, gop_size=4, streams=2)
    first = cenc.encode_gops(seqs, 1.0, 1.0, 1.0)
    last_recon = [first[k][2]["ref_frame"].clone() for k in range(2)]   # DPB tensors are views of recycled buffers
    for _ in range(2):
        again = cenc.encode_gops(seqs, 1.0, 1.0, 1.0)
        for k in range(2):
            assert [c[2] for c in again[k][0]] == [c[2] for c in first[k][0]], k
    for k in range(2):
        coded, bits, dpb = cenc.encoders[0].encode_gop(seqs[k], 1.0, 1.0, 1.0)
        assert [c[2] for c in coded] == [c[2] for c in first[k][0]] and bits == first[k][1], k
    recs = cenc.decode_gops([r[0] for r in first], 512, 768)
    for k in range(2):
        assert torch.equal(recs[k][-1], last_recon[k]), k
    for enc in cenc.encoders:
        enc.i_net.engine().release()
        enc.p_net.engine().release()


def test_config_c1_gop8_256_through_encode_decode_files(nets, tmp_path):
    """BASELINE configs[0] / SURVEY 8d C1: one 8-picture 256x256 GOP through the reference's own calls --
    IntraNoAR.encode_decode, then seven DMC.encode_decode with .bin files, q-scales 1.0 -- as
    video_coder.py:119-151 drives them.  The reference cannot write these files here (its rANS extension needs
    the absent ryg_rans header), so the call path is checked against what the reference DID produce
    for the same pictures: bits of every file within the README's "real bitstream vs estimate" margin of the
    estimate-path bits the reference computed (tests/golden/seq_256.npz holds pictures 0-2 of this sequence:
    seed 2; the README's "< 0.5 %" is for 1080p, at 256x256 the coder's fixed costs weigh more: 5 %), file headers, decode-from-file == encoder reconstruction, and the GOP recursion end to end."""
    from vcm_ts_amd import stream as S

    d, i = nets
    h = w = 256
    fx = golden("seq_256")
    fr = frames(2, 8, h, w)
    xs = [torch.from_numpy(fr[t : t + 1]).cuda() for t in range(8)]
    path = lambda t: os.path.join(tmp_path, f"im{t + 1:05d}.bin")
    r = i.encode_decode(xs[0], 1.0, path(0), pic_width=w, pic_height=h)
    assert r["bit"] == os.path.getsize(path(0)) * 8
    assert abs(r["bit"] - float(fx["i_bit"])) / float(fx["i_bit"]) < 0.05
    hh, ww, q_idx, payload = S.decode_i(path(0))
    assert (hh, ww, q_idx) == (h, w, 100)
    assert torch.equal(i.decompress(payload, h, w, 1.0)["x_hat"], r["x_hat"])
    # caller-owned copies throughout: the tensors the codec returns are views of its two alternating DPB buffer
    # sets, valid across the usual recursion but not across the extra estimate / decode calls made here
    own = lambda dd: {k: (None if v is None else v.clone()) for k, v in dd.items()}
    dpb = own({"ref_frame": r["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None})
    bits = [r["bit"]]
    for t in range(1, 8):
        prev = own(dpb)
        est = float(d.forward_one_frame(xs[t], prev, 1.0, 1.0)["bit"])  # entropy estimate on the same (decoded) DPB
        r = d.encode_decode(xs[t], dpb, path(t), pic_width=w, pic_height=h, mv_y_q_scale=1.0, y_q_scale=1.0)
        r["dpb"] = own(r["dpb"])
        assert r["bit"] == os.path.getsize(path(t)) * 8 and set(r) == {"dpb", "bit", "encoding_time", "decoding_time"}
        # (the fixture's P pictures are no yardstick here: the reference's estimate path feeds the UNclamped
        # reconstruction forward, video_model.py:535, the real coding loop the decoded, clamped one, :413)
        assert abs(r["bit"] - est) / est < 0.05, (t, r["bit"], est)
        mv_idx, y_idx, payload = S.decode_p(path(t))
        assert (mv_idx, y_idx) == (100, 100)
        dec = d.decompress(prev, payload, h, w, 1.0, 1.0)["dpb"]
        for k in dec:
            assert torch.equal(dec[k], r["dpb"][k]), (t, k)
        dpb = r["dpb"]
        bits.append(r["bit"])
    assert float(dpb["ref_frame"].min()) >= 0.0 and float(dpb["ref_frame"].max()) <= 1.0
    assert len(os.listdir(tmp_path)) == 8 and sum(bits) == sum(os.path.getsize(path(t)) for t in range(8)) * 8


def test_config_c5_four_rate_points_at_bench_size(nets):
    """BASELINE configs[4] / SURVEY 8d C5 at size: four rate points (the model's q-scale anchors) of a padded
    1080p picture pair in ONE batched compress call per picture (I, then P).  Each element's stream must decode
    with the ordinary batch-1 decoder at that rate to exactly the encoder's reconstruction for that element, and
    rates must be ordered (a larger q-scale codes fewer bits).  The detector perceptual loss of configs[4] is
    out of scope (pluggable callable)."""
    from vcm_ts_amd.pipeline import pad_frame

    d, i = nets
    fr = frames(12, 2, 1080, 1920)
    x0, x1 = (pad_frame(torch.from_numpy(fr[t : t + 1])).cuda() for t in range(2))
    qi = i.q_scale[:4].detach().reshape(-1)
    qmv, qy = d.mv_y_q_scale[:4].detach().reshape(-1), d.y_q_scale[:4].detach().reshape(-1)
    ci = i.compress(x0.expand(4, -1, -1, -1).contiguous(), qi.view(4, 1, 1, 1))
    assert len(ci["bit_streams"]) == 4
    rec_i = ci["x_hat"].clone()
    dpb = {"ref_frame": rec_i, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    cp = d.compress(x1.expand(4, -1, -1, -1).contiguous(), dpb, qmv.view(4, 1, 1, 1), qy.view(4, 1, 1, 1))
    assert len(cp["bit_streams"]) == 4
    rec_p = cp["dpb"]["ref_frame"].clone()
    sizes_i, sizes_p = [len(s) for s in ci["bit_streams"]], [len(s) for s in cp["bit_streams"]]
    for k in range(4):
        di = i.decompress(ci["bit_streams"][k], 1080, 1920, float(qi[k]))["x_hat"]
        assert torch.equal(di[0], rec_i[k]), k
        one = {"ref_frame": rec_i[k : k + 1].clone(), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        dp = d.decompress(one, cp["bit_streams"][k], 1080, 1920, float(qmv[k]), float(qy[k]))["dpb"]["ref_frame"]
        assert torch.equal(dp[0], rec_p[k]), k
    order = np.argsort(qi.cpu().numpy())
    assert all(sizes_i[a] >= sizes_i[b] for a, b in zip(order[:-1], order[1:])), (sizes_i, qi)
    d.engine().release()
    i.engine().release()
    torch.cuda.empty_cache()


def test_fast_mode_stays_within_tolerance_of_fp32_on_bench_content():
    """What bench.py prints as parity_mode_fp32.fast_vs_fp32, asserted: a GOP of the bench's own synthetic
    1080p content coded in split-fp16 mode and in exact-fp32 mode -- bits and GOP-mean PSNR within 1e-4
    (north_star's tolerance), no convolution output beyond the split-fp16 range (status word 0)."""
    from bench import GopQuality, synth_sequence
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.intra import IntraNoAR
    from vcm_ts_amd.pipeline import GopEncoder, pad_frame

    dev = torch.device("cuda:0")
    seq = [pad_frame(f) for f in synth_sequence(dev, 8, 1080, 1920, seed=0)]
    res = {}
    for prec in ("fp16x3", "fp32"):
        i, d = IntraNoAR(precision=prec).to(dev).eval(), DMC(precision=prec).to(dev).eval()
        i.engine().range_check = d.engine().range_check = True
        q = GopQuality(seq, 1080, 1920)
        _, bits, _ = GopEncoder(i, d, gop_size=8).encode_gop(seq, 1.0, 1.0, 1.0, on_recon=q)
        res[prec] = (bits, q.psnr(), i.engine().read_status() | d.engine().read_status())
        i.engine().release()
        d.engine().release()
        del i, d
        torch.cuda.empty_cache()
    (bf, pf, sf), (b32, p32, _) = res["fp16x3"], res["fp32"]
    assert sf == 0
    assert abs(bf - b32) <= 1e-4 * b32, (bf, b32)
    assert abs(pf.mean() - p32.mean()) <= 1e-4 * max(abs(p32.mean()), 1.0), (pf.mean(), p32.mean())
    assert np.abs(pf - p32).max() < 5e-3  # per picture, in dB
