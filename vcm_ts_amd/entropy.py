"""Host side of the entropy model: the rANS coder objects and the CDF-table builders.

Mirrors the surface the reference's L2 code uses
(/root/reference/DCVC_HEM/src/entropy_models/entropy_models.py):

* ``BufferedRansEncoder`` / ``RansDecoder`` -- same methods and argument meaning as the
  pybind classes of ``MLCodec_rans`` (rans_interface.cpp:246-261), implemented over the
  C ABI of include/dcvc_rans.h;
* ``pmf_to_quantized_cdf`` -- ``MLCodec_CXX.pmf_to_quantized_cdf`` (ops.cpp:84-91);
* ``EntropyCoder`` -- entropy_models.py:9-51;
* ``scale_table_cdfs`` / ``factorized_cdfs`` / ``factorized_param_block`` -- what
  ``GaussianEncoder.update`` (:224-262) and ``BitEstimator.update`` (:119-174) build once
  per model before real coding.  They run on the host (one-off, ~400 short rows) with the
  same fp32 torch-CPU arithmetic as the reference so the integer tables are identical.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import lib


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.c_void_p)


class RansError(RuntimeError):
    pass


def _chk(code, what):
    if code < 0:
        raise RansError(f"{what}: status {code}")
    return code


class BufferedRansEncoder:
    def __init__(self):
        self._L = lib.rans()
        self._h = self._L.dcvc_rans_encoder_create()
        if not self._h:
            raise MemoryError("dcvc_rans_encoder_create")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.dcvc_rans_encoder_destroy(self._h)
            self._h = None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        s, sp = _i32(symbols)
        i, ip = _i32(indexes)
        if s.size != i.size:
            raise ValueError("symbols and indexes differ in length")
        c, cp = _i32(cdfs)
        z, zp = _i32(cdfs_sizes)
        o, op = _i32(offsets)
        _chk(self._L.dcvc_rans_encoder_encode_with_indexes(self._h, sp, ip, s.size, cp, c.shape[0], c.shape[1], zp, op),
             "encode_with_indexes")

    def flush(self) -> bytes:
        cap = self._L.dcvc_rans_encoder_flush_bound(self._h)
        buf = (C.c_uint8 * cap)()
        n = _chk(self._L.dcvc_rans_encoder_flush(self._h, buf, cap), "flush")
        return bytes(memoryview(buf)[:n])

    def reset(self):
        self._L.dcvc_rans_encoder_reset(self._h)


class RansDecoder:
    def __init__(self):
        self._L = lib.rans()
        self._h = self._L.dcvc_rans_decoder_create()
        if not self._h:
            raise MemoryError("dcvc_rans_decoder_create")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.dcvc_rans_decoder_destroy(self._h)
            self._h = None

    def set_stream(self, stream: bytes):
        _chk(self._L.dcvc_rans_decoder_set_stream(self._h, stream, len(stream)), "set_stream")

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        i, ip = _i32(indexes)
        c, cp = _i32(cdfs)
        z, zp = _i32(cdfs_sizes)
        o, op = _i32(offsets)
        out = np.empty(i.size, np.int32)
        _chk(self._L.dcvc_rans_decoder_decode_stream(self._h, ip, i.size, cp, c.shape[0], c.shape[1], zp, op,
                                                     out.ctypes.data_as(C.c_void_p)), "decode_stream")
        return out


def pmf_to_quantized_cdf(pmf, precision: int = 16):
    p = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.empty(p.size + 1, np.uint32)
    _chk(lib.rans().dcvc_pmf_to_quantized_cdf(p.ctypes.data_as(C.c_void_p), p.size, precision,
                                              out.ctypes.data_as(C.c_void_p)), "pmf_to_quantized_cdf")
    return out.tolist()


class EntropyCoder:
    """entropy_models.py:9-51."""

    def __init__(self):
        self.encoder = BufferedRansEncoder()
        self.decoder = RansDecoder()

    @staticmethod
    def pmf_to_quantized_cdf(pmf, precision=16):
        return torch.IntTensor(pmf_to_quantized_cdf(pmf.tolist() if hasattr(pmf, "tolist") else pmf, precision))

    @staticmethod
    def pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i in range(len(pmf_length)):
            row = torch.cat((pmf[i, : int(pmf_length[i])], tail_mass[i]), dim=0)
            q = EntropyCoder.pmf_to_quantized_cdf(row, 16)
            cdf[i, : q.size(0)] = q
        return cdf

    def set_stream(self, stream):
        self.decoder.set_stream(stream)

    def encode_with_indexes(self, symbols_list, indexes_list, cdf, cdf_length, offset):
        self.encoder.encode_with_indexes(symbols_list, indexes_list, cdf, cdf_length, offset)

    def flush_encoder(self):
        return self.encoder.flush()

    def reset_encoder(self):
        self.encoder.reset()

    def decode_stream(self, indexes, cdf, cdf_length, offset):
        rv = self.decoder.decode_stream(indexes, cdf, cdf_length, offset)
        return torch.from_numpy(rv.astype(np.float32)).reshape(1, -1, 1, 1)


# ----------------------------------------------------------------------------- table builders
SCALE_LEVELS = 256
SCALE_MAX = 64.0


def scale_log_params(distribution: str):
    smin = 0.01 if distribution == "laplace" else 0.11
    lmin = math.log(smin)
    return lmin, (math.log(SCALE_MAX) - lmin) / (SCALE_LEVELS - 1)


def _dist(distribution, scale):
    loc = torch.zeros_like(scale)
    if distribution == "laplace":
        return torch.distributions.laplace.Laplace(loc, scale)
    return torch.distributions.normal.Normal(loc, scale)


def scale_table_cdfs(distribution: str = "laplace"):
    """(cdf int32 (256, L), lengths int32 (256), offsets int32 (256)) of the 256-level
    log-spaced scale table (entropy_models.py:212-262)."""
    lmin, _ = scale_log_params(distribution)
    table = torch.exp(torch.linspace(lmin, math.log(SCALE_MAX), SCALE_LEVELS))
    d = _dist(distribution, table)
    center = torch.full_like(table, 50.0)
    for i in range(50, 1, -1):  # ends at the SMALLEST i in [2, 50] whose tail is below 1e-4
        center = torch.where(d.cdf(torch.full_like(table, float(i))) > 0.9999, torch.full_like(table, float(i)), center)
    center = center.int()
    lengths = 2 * center + 1
    max_len = int(lengths.max())
    samples = (torch.arange(max_len) - center[:, None]).float()
    d = _dist(distribution, torch.zeros_like(samples) + table[:, None])
    upper, lower = d.cdf(samples + 0.5), d.cdf(samples - 0.5)
    cdf = EntropyCoder.pmf_to_cdf(upper - lower, 2 * lower[:, :1], lengths, max_len)
    return cdf.numpy(), (lengths + 2).int().numpy(), (-center).int().numpy()


def _factorized_cdf(p, x):
    """sigmoid(f4(f3(f2(f1(x))))) with per-channel parameters (entropy_models.py:68-73,109-117).
    p: dict h1..h4, b1..b4, a1..a3 of (1, ch, 1, 1) tensors; x broadcastable to (1, ch, 1, K)."""
    for i in (1, 2, 3):
        x = x * torch.nn.functional.softplus(p[f"h{i}"]) + p[f"b{i}"]
        x = x + torch.tanh(x) * torch.tanh(p[f"a{i}"])
    x = x * torch.nn.functional.softplus(p["h4"]) + p["b4"]
    return torch.sigmoid(x)


def factorized_cdfs(p):
    """Per-channel tables of the factorised prior (entropy_models.py:119-174)."""
    ch = p["h1"].shape[1]
    zero = torch.zeros(ch)
    lo, hi = zero + 50, zero + 50
    for i in range(50, 1, -1):
        pr = _factorized_cdf(p, (zero - i)[None, :, None, None]).squeeze()
        lo = torch.where(pr < 0.0001, zero + i, lo)
    for i in range(50, 1, -1):
        pr = _factorized_cdf(p, (zero + i)[None, :, None, None]).squeeze()
        hi = torch.where(pr > 0.9999, zero + i, hi)
    lo, hi = lo.int(), hi.int()
    lengths = hi + lo + 1
    max_len = int(lengths.max())
    samples = torch.arange(max_len)[None, :] + (zero - lo)[:, None, None]  # (ch, 1, K)
    lower = _factorized_cdf(p, samples - 0.5).squeeze(0)
    upper = _factorized_cdf(p, samples + 0.5).squeeze(0)
    pmf = (upper - lower)[:, 0, :]
    tail = lower[:, 0, :1] + (1.0 - upper[:, 0, -1:])
    cdf = EntropyCoder.pmf_to_cdf(pmf, tail, lengths, max_len)
    return cdf.numpy(), (lengths + 2).int().numpy(), (-lo).int().numpy()


def factorized_params(sd, prefix):
    """Collect h/b/a of the four layers from a state dict with the reference's key names."""
    p = {}
    for i in (1, 2, 3, 4):
        p[f"h{i}"] = sd[f"{prefix}.f{i}.h"].detach().float().cpu()
        p[f"b{i}"] = sd[f"{prefix}.f{i}.b"].detach().float().cpu()
        if i != 4:
            p[f"a{i}"] = sd[f"{prefix}.f{i}.a"].detach().float().cpu()
    return p


def factorized_param_block(p):
    """(11, ch) fp32 block in the order dcvc_factorized_bits expects."""
    rows = []
    for i in (1, 2, 3):
        rows += [p[f"h{i}"], p[f"b{i}"], p[f"a{i}"]]
    rows += [p["h4"], p["b4"]]
    return torch.stack([r.reshape(-1) for r in rows]).contiguous()
