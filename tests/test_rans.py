"""Entropy-coder boundary (include/dcvc_rans.h): product C++ coder vs the oracle's C and
pure-Python restatements, golden byte strings, tables vs reference-generated fixtures."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from oracle import rans_py
from tests.util import ROOT, golden, oracle_weights
from vcm_ts_amd import entropy as E
from vcm_ts_amd import lib


def _oracle_c():
    path = os.path.join(ROOT, "oracle", "liboracle_ref.so")
    if not os.path.exists(path):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    L = C.CDLL(path)
    L.ref_enc_new.restype = C.c_void_p
    L.ref_enc_free.argtypes = [C.c_void_p]
    L.ref_enc_encode_with_indexes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p]
    L.ref_enc_flush.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.ref_enc_flush.restype = C.c_void_p
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_dec_new.restype = C.c_void_p
    L.ref_dec_free.argtypes = [C.c_void_p]
    L.ref_dec_set_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.ref_dec_decode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    L.ref_pmf_to_quantized_cdf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return L


def oracle_c_encode(batches):
    L = _oracle_c()
    e = L.ref_enc_new()
    keep = []
    for sym, idx, cdf, ln, off in batches:
        arrs = [np.ascontiguousarray(a, np.int32) for a in (sym, idx, cdf, ln, off)]
        keep.append(arrs)
        s, i, c, l, o = arrs
        L.ref_enc_encode_with_indexes(e, s.ctypes.data, i.ctypes.data, s.size, c.ctypes.data, c.shape[1],
                                      l.ctypes.data, o.ctypes.data)
    n = C.c_size_t()
    p = L.ref_enc_flush(e, C.byref(n))
    data = C.string_at(p, n.value)
    L.ref_free(p)
    L.ref_enc_free(e)
    return data


def oracle_c_decode(data, idx_batches, cdf, ln, off):
    L = _oracle_c()
    d = L.ref_dec_new()
    L.ref_dec_set_stream(d, data, len(data))
    c, l, o = (np.ascontiguousarray(a, np.int32) for a in (cdf, ln, off))
    outs = []
    for idx in idx_batches:
        i = np.ascontiguousarray(idx, np.int32)
        out = np.empty(i.size, np.int32)
        L.ref_dec_decode_stream(d, i.ctypes.data, i.size, c.ctypes.data, c.shape[1], l.ctypes.data, o.ctypes.data,
                                out.ctypes.data)
        outs.append(out)
    L.ref_dec_free(d)
    return outs


def product_encode(batches):
    enc = E.BufferedRansEncoder()
    enc.reset()
    for b in batches:
        enc.encode_with_indexes(*b)
    return enc.flush()


@pytest.fixture(scope="module")
def laplace_tables():
    t = golden("tables")
    return t["dmc_scale_cdf"], t["dmc_scale_len"], t["dmc_scale_off"]


def test_library_exports_every_declared_symbol():
    R = lib.rans()
    for s in lib.RANS_SYMBOLS:
        assert hasattr(R, s), s
    H = lib.hip()  # loads without a GPU; no compute call here
    for s in lib.HIP_SYMBOLS:
        assert hasattr(H, s), s
    # every function declared in the headers is bound
    import re

    for hdrs, syms in ((("dcvc_rans.h",), lib.RANS_SYMBOLS), (("dcvc_hip.h", "dcvc_hip_grad.h", "dcvc_hip_rans.h"), lib.HIP_SYMBOLS)):
        text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in hdrs)
        declared = set(re.findall(r"\b(dcvc_[a-z0-9_]+)\s*\(", text))
        assert declared == set(syms), declared ^ set(syms)


def test_golden_byte_strings(laplace_tables):
    cdf, ln, off = laplace_tables
    fx = golden("rans_bytes")
    for name in ("small", "bypass", "huge", "empty", "one"):
        sym, idx, want = fx[name + "_sym"], fx[name + "_idx"], fx[name + "_bytes"].tobytes()
        assert product_encode([(sym, idx, cdf, ln, off)]) == want, name
        assert oracle_c_encode([(sym, idx, cdf, ln, off)]) == want, name
        dec = E.RansDecoder()
        dec.set_stream(want)
        np.testing.assert_array_equal(dec.decode_stream(idx, cdf, ln, off), sym)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_three_implementations_agree_on_random_streams(laplace_tables, seed):
    cdf, ln, off = laplace_tables
    g = np.random.default_rng(seed)
    batches = []
    for n, spread in ((500, 3), (300, 60), (50, 5000), (1, 1)):
        idx = g.integers(0, 256, n).astype(np.int32)
        sym = np.rint(g.laplace(0, spread, n)).astype(np.int32)
        batches.append((sym, idx, cdf, ln, off))
    a = product_encode(batches)
    assert a == oracle_c_encode(batches)
    assert a == rans_py.encode([(s.tolist(), i.tolist(), c, l, o) for s, i, c, l, o in batches])
    dec = E.RansDecoder()
    dec.set_stream(a)
    py = rans_py.Decoder(a)
    ref = oracle_c_decode(a, [b[1] for b in batches], cdf, ln, off)
    for (sym, idx, *_), r in zip(batches, ref):  # shared cursor over successive calls
        np.testing.assert_array_equal(dec.decode_stream(idx, cdf, ln, off), sym)
        np.testing.assert_array_equal(r, sym)
        assert py.decode(idx.tolist(), cdf, ln, off) == sym.tolist()


def test_reference_symbol_planes_round_trip(laplace_tables):
    """The six planes of a real P-frame (symbols and indexes produced by the reference
    itself, tests/golden/seq_256.npz) through one stream, in bitstream order."""
    cdf, ln, off = laplace_tables
    t = golden("tables")
    fx = golden("seq_256")
    z = (t["dmc_z_cdf"], t["dmc_z_len"], t["dmc_z_off"])
    zmv = (t["dmc_zmv_cdf"], t["dmc_zmv_len"], t["dmc_zmv_off"])

    def chan_idx(sym):
        n, c, h, w = sym.shape
        return np.broadcast_to(np.arange(c, dtype=np.int32)[None, :, None, None], sym.shape).reshape(-1)

    order = [("mv_z", zmv), ("mv_y0", None), ("mv_y1", None), ("z", z), ("y0", None), ("y1", None)]
    batches = []
    for tag, tab in order:
        sym = fx["p1_sym_" + tag].astype(np.int32)
        if tab is None:
            batches.append((sym.reshape(-1), fx["p1_idx_" + tag].astype(np.int32).reshape(-1), cdf, ln, off))
        else:
            batches.append((sym.reshape(-1), chan_idx(sym), *tab))
    data = product_encode(batches)
    assert data == oracle_c_encode(batches)
    dec = E.RansDecoder()
    dec.set_stream(data)
    for sym, idx, c, l, o in batches:
        np.testing.assert_array_equal(dec.decode_stream(idx, c, l, o), sym)
    nsym = sum(b[0].size for b in batches)
    assert 0 < len(data) < 4 * nsym


def test_error_codes(laplace_tables):
    cdf, ln, off = laplace_tables
    enc = E.BufferedRansEncoder()
    with pytest.raises(E.RansError):
        enc.encode_with_indexes(np.zeros(3, np.int32), np.array([0, 256, 1], np.int32), cdf, ln, off)
    assert enc.flush() == product_encode([])  # failed call left nothing behind
    with pytest.raises(ValueError):
        enc.encode_with_indexes(np.zeros(3, np.int32), np.zeros(2, np.int32), cdf, ln, off)
    dec = E.RansDecoder()
    with pytest.raises(E.RansError):
        dec.decode_stream(np.zeros(1, np.int32), cdf, ln, off)  # no stream
    data = product_encode([(np.arange(-40, 40, dtype=np.int32), np.full(80, 200, np.int32), cdf, ln, off)])
    dec.set_stream(data[: len(data) // 2])
    with pytest.raises(E.RansError):
        dec.decode_stream(np.full(80, 200, np.int32), cdf, ln, off)
    with pytest.raises(E.RansError):
        dec.set_stream(b"abc")


def test_null_handles_and_buffers_are_refused():
    """The raw C ABI (include/dcvc_rans.h) with NULL handles / buffers: a status code, never a dereference."""
    from vcm_ts_amd import lib

    L = lib.rans()
    assert L.dcvc_rans_encoder_reset(None) == -1
    assert L.dcvc_rans_encoder_encode_with_indexes(None, None, None, 4, None, 1, 1, None, None) == -1
    assert L.dcvc_rans_encoder_flush_bound(None) == -1
    assert L.dcvc_rans_encoder_flush(None, None, 0) == -1
    assert L.dcvc_rans_decoder_set_stream(None, None, 0) == -1
    assert L.dcvc_rans_decoder_decode_stream(None, None, 4, None, 1, 1, None, None, None) == -1
    assert L.dcvc_pmf_to_quantized_cdf(None, 0, 16, None) == -1
    L.dcvc_rans_encoder_destroy(None)  # (destroying nothing is a no-op)
    L.dcvc_rans_decoder_destroy(None)


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_decoder_is_memory_safe_on_corrupt_streams_under_asan(tmp_path):
    """tests/fuzz/rans_fuzz.cpp: the product coder (vcm_ts_amd/csrc/rans.cpp) built with AddressSanitizer + UBSan; random
    tables and planes (in-table and escape-coded) round-trip, and truncated / bit-flipped / garbage-extended / escape-inflated
    payloads are answered with a status code or other symbols -- never with an out-of-bounds access (any sanitizer report
    aborts the run).  A payload comes from a file: the reference's decoder only asserts (rans_interface.cpp:184-244)."""
    exe = tmp_path / "rans_fuzz"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "vcm_ts_amd", "csrc", "rans.cpp"),
                            os.path.join(ROOT, "tests", "fuzz", "rans_fuzz.cpp"), "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    for seed in (1, 2, 3):
        run = subprocess.run([str(exe), "1500", str(seed)], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, (seed, run.stdout[-500:], run.stderr[-3000:])
        assert "rans_fuzz: 1500 rounds" in run.stdout and "refused with a status" in run.stdout


def test_corrupt_escape_count_is_rejected(laplace_tables):
    """A stream whose escape announces more than 8 raw nibbles cannot come from any encoder (a 32-bit
    value has 8): the product decoder must return DCVC_RANS_E_STREAM instead of shifting an int32 by
    >= 32 bits (rans.cpp), and the Python oracle must refuse it the same way.  The stream is built by
    the oracle's record writer with a forged count nibble."""
    cdf, ln, off = laplace_tables
    row = 3
    sentinel = int(ln[row]) - 2
    start, freq = int(cdf[row][sentinel]), int(cdf[row][sentinel + 1]) - int(cdf[row][sentinel])
    # decode order: sentinel symbol, count nibble 12, then 12 raw nibbles
    recs = [(start, freq, False), (12, 0, True)] + [(5, 0, True)] * 12
    data = rans_py.flush_records(recs)
    dec = E.RansDecoder()
    dec.set_stream(data)
    with pytest.raises(E.RansError):
        dec.decode_stream(np.array([row], np.int32), cdf, ln, off)
    with pytest.raises(ValueError):
        rans_py.Decoder(data).decode([row], cdf, ln, off)
    # control: the same construction with a legal count decodes in both
    raw = 2 * 7  # value = sentinel + 7
    recs = [(start, freq, False), (1, 0, True), (raw, 0, True)]
    data = rans_py.flush_records(recs)
    dec.set_stream(data)
    want = sentinel + 7 + int(off[row])
    assert dec.decode_stream(np.array([row], np.int32), cdf, ln, off).tolist() == [want]
    assert rans_py.Decoder(data).decode([row], cdf, ln, off) == [want]


def test_flush_resets_like_reset(laplace_tables):
    cdf, ln, off = laplace_tables
    enc = E.BufferedRansEncoder()
    enc.encode_with_indexes(np.array([1, 2, 3], np.int32), np.array([5, 6, 7], np.int32), cdf, ln, off)
    a = enc.flush()
    enc.reset()
    enc.encode_with_indexes(np.array([1, 2, 3], np.int32), np.array([5, 6, 7], np.int32), cdf, ln, off)
    assert enc.flush() == a


def test_pmf_to_quantized_cdf_matches_reference_build():
    fx = golden("tables")
    L = _oracle_c()
    for k in range(int(fx["n_pmf"])):
        pmf = fx[f"pmf_{k}"]
        want = fx[f"qcdf_{k}"]
        np.testing.assert_array_equal(np.array(E.pmf_to_quantized_cdf(pmf), np.int64), want)
        out = np.empty(pmf.size + 1, np.uint32)
        p = np.ascontiguousarray(pmf, np.float32)
        L.ref_pmf_to_quantized_cdf(p.ctypes.data, p.size, 16, out.ctypes.data)
        np.testing.assert_array_equal(out.astype(np.int64), want)


def test_tables_match_reference():
    fx = golden("tables")
    for tag, dist in (("dmc", "laplace"), ("intra", "gaussian")):
        c, l, o = E.scale_table_cdfs(dist)
        np.testing.assert_array_equal(c, fx[f"{tag}_scale_cdf"])
        np.testing.assert_array_equal(l, fx[f"{tag}_scale_len"])
        np.testing.assert_array_equal(o, fx[f"{tag}_scale_off"])
    for tag, kind, name in (("dmc_z", "dmc", "bit_estimator_z"), ("dmc_zmv", "dmc", "bit_estimator_z_mv"),
                            ("intra_z", "intra", "bit_estimator_z")):
        c, l, o = E.factorized_cdfs(E.factorized_params(oracle_weights(kind), name))
        np.testing.assert_array_equal(c, fx[f"{tag}_cdf"])
        np.testing.assert_array_equal(l, fx[f"{tag}_len"])
        np.testing.assert_array_equal(o, fx[f"{tag}_off"])


def test_device_format_oracle_round_trip_and_lane_streams(laplace_tables):
    """oracle/drans_py.py (restatement of the opt-in device format, include/dcvc_hip_rans.h): a
    section's lane streams are ordinary single-stream rANS payloads of the product's host coder, and
    the section decodes back; the lane-count policy matches the library's."""
    from oracle import drans_py as D

    cdf, ln, off = laplace_tables
    rng = np.random.default_rng(11)
    n, L = 700, 64
    idx = rng.integers(0, 256, n).astype(np.int32)
    sym = np.rint(rng.laplace(0, 5, n)).astype(np.int32)
    sym[::41] = 500
    sym[3] = -70000
    sec = D.encode_section(sym, idx, cdf, ln, off, lanes=L)
    hdr = np.frombuffer(sec[: 8 + 4 * L], np.uint32)
    assert hdr[0] == n and hdr[1] == L
    pos = 8 + 4 * L
    for j in range(L):
        want = product_encode([(sym[j::L], idx[j::L], cdf, ln, off)])
        assert sec[pos : pos + len(want)] == want and hdr[2 + j] * 4 == len(want), j
        pos += len(want)
    assert pos == len(sec)
    back, end = D.decode_section(D.MAGIC + sec, 4, idx, cdf, ln, off)
    np.testing.assert_array_equal(back, sym)
    assert end == 4 + len(sec)
    for m in (1, 63, 512, 513, 32640, 391680, 10**7):
        assert D.default_lanes(m) == lib.hip().dcvc_drans_default_lanes(m), m


def test_index_bin_edges_reproduce_build_indexes():
    """entropy.scale_index_edges (the 255 fp32 bin edges the kernels count instead of taking a logarithm):
    counting edges <= s reproduces the reference's build_indexes on its own sweep (tables.npz), on every float
    within 3 ulp of every edge and on 2 M random scales -- on the CPU, the same search the kernels perform."""
    t = golden("tables")
    s = torch.from_numpy(t["idx_sweep_in"])
    for dist, key in (("laplace", "idx_sweep_laplace"), ("gaussian", "idx_sweep_gauss")):
        edges = E.scale_index_edges(dist)
        assert edges.shape == (256,) and torch.isinf(edges[-1]) and bool((edges[1:] > edges[:-1]).all())
        # the product path's edges are constants (vcm_ts_amd/index_edges.py): this host's torch-CPU logf must agree
        # with them bit for bit, or its streams would not decode elsewhere
        assert torch.equal(E.derive_scale_index_edges(dist).view(torch.int32), edges.view(torch.int32))
        count = lambda v: (edges[None, :] <= v[:, None]).sum(1).int()
        np.testing.assert_array_equal(count(s).numpy(), t[key])
        eb = edges[:-1].view(torch.int32)
        near = torch.cat([(eb + d).view(torch.float32) for d in range(-3, 4)])
        assert torch.equal(count(near), E.reference_scale_indexes(near.clone(), dist))
        rnd = torch.exp(torch.empty(2_000_000).uniform_(-13.0, 6.0, generator=torch.Generator().manual_seed(1)))
        assert torch.equal(torch.searchsorted(edges, rnd, right=True).int(), E.reference_scale_indexes(rnd.clone(), dist))
