"""Geometry helpers and `.bin` wire format (vcm_ts_amd/stream.py) against values and bytes
produced by the reference's own stream_helper.py (tests/golden/stream.npz)."""
import os

import numpy as np
import pytest

from tests.util import golden
from vcm_ts_amd import stream as S


def test_geometry_and_q_rounding_match_reference():
    fx = golden("stream")
    for (h, w), pad, d16, d64 in zip(fx["sizes"], fx["padding"], fx["down16"], fx["down64"]):
        assert S.get_padding_size(int(h), int(w)) == tuple(int(v) for v in pad)
        assert S.get_downsampled_shape(int(h), int(w), 16) == tuple(int(v) for v in d16)
        assert S.get_downsampled_shape(int(h), int(w), 64) == tuple(int(v) for v in d64)
    for q, qs, qi in zip(fx["q_in"], fx["q_scale"], fx["q_index"]):
        got_s, got_i = S.get_rounded_q(float(q))
        assert got_i == int(qi) and got_s == pytest.approx(float(qs), abs=0)


def test_bin_files_are_byte_identical_to_reference(tmp_path):
    fx = golden("stream")
    payload = fx["payload"].tobytes()
    p = os.path.join(tmp_path, "p.bin")
    S.encode_p(payload, 123, 45678, p)
    assert open(p, "rb").read() == fx["bin_p"].tobytes()
    assert S.decode_p(p) == (123, 45678, payload)
    S.encode_i(1080, 1920, 150, payload, p)
    assert open(p, "rb").read() == fx["bin_i"].tobytes()
    assert S.decode_i(p) == (1080, 1920, 150, payload)
    assert S.filesize(p) == len(fx["bin_i"])
    # empty payload and error path
    S.encode_p(b"", 1, 2, p)
    assert S.decode_p(p) == (1, 2, b"")
    with pytest.raises(ValueError):
        S.filesize(os.path.join(tmp_path, "missing.bin"))


def test_pad_frame_matches_reference_padding():
    import torch

    from vcm_ts_amd.pipeline import pad_frame

    x = torch.rand(1, 3, 1080, 1920)
    y = pad_frame(x)
    assert y.shape == (1, 3, 1088, 1920)
    assert torch.equal(y[..., :1080, :], x) and float(y[..., 1080:, :].abs().max()) == 0.0


def test_rate_point_selection_matches_reference_interpolation():
    """run_codec --rate-count / --quality (video_coder.py:181-197): interpolate_log against values the reference's
    own function produced (tests/golden/make_golden_rate_points.py), and the anchor order of the q_scale tensors."""
    from vcm_ts_amd.run_codec import interpolate_log, rate_point_q_scales

    fx = golden("rate_points")
    for k, (lo, hi, n) in enumerate(zip(fx["lo"], fx["hi"], fx["num"])):
        np.testing.assert_array_equal(interpolate_log(float(lo), float(hi), int(n)), fx[f"dec_{k}"])
        np.testing.assert_array_equal(interpolate_log(float(lo), float(hi), int(n), decending=False), fx[f"asc_{k}"])
    # checkpoints store the anchors coarsest first: [0] is the maximum, [-1] the minimum (video_coder.py:183-184)
    i_q, y_q, mv_q = np.array([1.8, 1.2, 0.8, 0.5]), np.array([2.7, 1.5, 0.9, 0.3]), np.array([12.0, 3.0, 1.0, 0.05])
    q = rate_point_q_scales(i_q, y_q, mv_q, 6, 1)
    assert q == (float(fx["dec_0"][1]), float(interpolate_log(0.05, 12.0, 6)[1]), float(interpolate_log(0.3, 2.7, 6)[1]))
    assert rate_point_q_scales(i_q, y_q, mv_q, 6, 0) == (1.8, 12.0, 2.7)
    with pytest.raises(ValueError):
        rate_point_q_scales(i_q, y_q, mv_q, 6, 6)


def test_prefetching_png_reader_returns_the_frames_of_the_sequential_reader(tmp_path):
    """run_codec.PNGReader.prefetching (worker threads decoding ahead) yields the same arrays in the same order as
    read_one_frame (DCVC_HEM/src/utils/png_reader.py:10-46: imNNNNN.png, RGB float32 / 255), for both naming widths,
    and stops at the first missing index."""
    from PIL import Image

    import torch

    from vcm_ts_amd.run_codec import PNGReader, u8_to_unit_float

    rng = np.random.default_rng(5)
    for width, n in ((5, 7), (1, 3)):
        d = tmp_path / f"w{width}"
        d.mkdir()
        for t in range(n):
            Image.fromarray(rng.integers(0, 256, (20, 36, 3), dtype=np.uint8)).save(d / f"im{str(t + 1).zfill(width)}.png")
        Image.fromarray(np.zeros((20, 36, 3), np.uint8)).save(d / f"im{str(n + 2).zfill(width)}.png")  # after a gap: never read
        seq, r = [], PNGReader(str(d))
        while (f := r.read_one_frame()) is not None:
            seq.append(f)
        pre = list(PNGReader(str(d)).prefetching(workers=3, depth=4))
        raw = list(PNGReader(str(d)).prefetching(workers=3, depth=4, raw=True))
        assert len(seq) == len(pre) == len(raw) == n
        for a, b, u in zip(seq, pre, raw):
            assert a.dtype == np.float32 and a.shape == (3, 20, 36)
            np.testing.assert_array_equal(a, b)
            # the uint8 route (4x fewer bytes to the device, table look-up there) gives the same floats, bit for bit
            np.testing.assert_array_equal(a, u8_to_unit_float(torch.from_numpy(u)).numpy()[0])
    every = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    np.testing.assert_array_equal(u8_to_unit_float(torch.from_numpy(every)).numpy()[0], every.astype("float32").transpose(2, 0, 1) / 255.0)


def test_png_writer_pool_propagates_failures_and_bounds_its_queue(tmp_path):
    """run_codec.PNGWriters: a failed PNG write surfaces in the caller instead of vanishing with its
    future, at most 2 x workers pictures wait in memory, workers = 0 writes inline."""
    import numpy as np
    import pytest

    from vcm_ts_amd.run_codec import PNGWriters

    a = np.zeros((8, 8, 3), np.float32)
    with PNGWriters(2) as w:
        for k in range(9):
            w.submit(a, str(tmp_path / f"ok{k}.png"))
            assert len(w.pending) <= 4
    assert sorted(p.name for p in tmp_path.iterdir()) == [f"ok{k}.png" for k in range(9)]
    with pytest.raises(OSError):
        with PNGWriters(2) as w:
            w.submit(a, str(tmp_path / "no_such_dir" / "x.png"))
    with pytest.raises(OSError):
        PNGWriters(0).submit(a, str(tmp_path / "no_such_dir" / "x.png"))  # inline
    w0 = PNGWriters(0)
    w0.submit(a, str(tmp_path / "inline.png"))
    assert w0.pool is None and (tmp_path / "inline.png").exists()


def test_truncated_bin_files_are_refused_by_name(tmp_path):
    """A .bin file shorter than its own header announces (an interrupted copy) is refused when it is read -- the
    reference's f.read(n) returns what is left and the entropy decoder runs off the end."""
    from vcm_ts_amd import stream as S

    pi, pp = str(tmp_path / "i.bin"), str(tmp_path / "p.bin")
    S.encode_i(1080, 1920, 37, bytes(range(200)), pi)
    S.encode_p(bytes(range(100)), 12, 34, pp)
    assert S.decode_i(pi) == (1080, 1920, 37, bytes(range(200))) and S.decode_p(pp) == (12, 34, bytes(range(100)))
    for path, fn, keep in ((pi, S.decode_i, 150), (pp, S.decode_p, 50), (pi, S.decode_i, 5), (pp, S.decode_p, 3), (pi, S.decode_i, 0)):
        data = open(path, "rb").read()
        cut = str(tmp_path / "cut.bin")
        open(cut, "wb").write(data[:keep])
        with pytest.raises(ValueError, match="truncated"):
            fn(cut)
