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


def reference_scale_indexes(scales: torch.Tensor, distribution: str) -> torch.Tensor:
    """GaussianEncoder.build_indexes exactly as the reference evaluates it (entropy_models.py:264-268:
    fp32 torch-CPU tensor ops, Python-double constants).  Host-side; used to derive the bin edges."""
    lmin, lstep = scale_log_params(distribution)
    scales = torch.maximum(scales, torch.zeros_like(scales) + 1e-5)
    indexes = (torch.log(scales) - lmin) / lstep
    return indexes.clamp_(0, SCALE_LEVELS - 1).int()


_EDGE_CACHE = {}


def scale_index_edges(distribution: str) -> torch.Tensor:
    """(256,) fp32: edge[k-1] = the smallest fp32 scale whose reference index is >= k (k = 1..255),
    edge[255] = +inf, so that index(s) = #{edges <= s} reproduces build_indexes bit for bit without a
    device logarithm (the kernels' logf differs from torch-CPU's by an ulp next to a bin edge).
    The values are CONSTANTS (vcm_ts_amd/index_edges.py): every host bins alike, whatever its libm;
    derive_scale_index_edges() is how they were found and how a test re-checks them."""
    if distribution not in _EDGE_CACHE:
        from . import index_edges

        bits = {"laplace": index_edges.LAPLACE_EDGE_BITS, "gaussian": index_edges.GAUSSIAN_EDGE_BITS}[distribution]
        assert len(bits) == SCALE_LEVELS - 1
        edges = torch.tensor(list(bits) + [0x7F800000], dtype=torch.int32).view(torch.float32).clone()
        _EDGE_CACHE[distribution] = edges
    return _EDGE_CACHE[distribution]


def derive_scale_index_edges(distribution: str) -> torch.Tensor:
    """The edges of scale_index_edges() re-derived on this host: bisection over the float's bit pattern -- positive
    floats order like their int32 bits -- with the reference formula itself (reference_scale_indexes: torch-CPU
    logf), all 255 edges at once; monotonicity checked on both sides of every edge."""
    k = torch.arange(1, SCALE_LEVELS + 1, dtype=torch.int32)          # 256 lanes; the last is a dummy
    lo = torch.full((SCALE_LEVELS,), 1e-5, dtype=torch.float32).view(torch.int32).clone()    # index 0 < k
    hi = torch.full((SCALE_LEVELS,), 128.0, dtype=torch.float32).view(torch.int32).clone()   # index 255
    for _ in range(32):
        mid = lo + (hi - lo) // 2
        ge = reference_scale_indexes(mid.view(torch.float32).clone(), distribution) >= k
        hi = torch.where(ge, mid, hi)
        lo = torch.where(ge, lo, mid)
    edges = hi.view(torch.float32).clone()
    edges[SCALE_LEVELS - 1] = float("inf")
    below = (hi - 1).view(torch.float32).clone()
    kk = k[:-1]
    if not (torch.equal(reference_scale_indexes(edges[:-1].clone(), distribution), kk)
            and torch.equal(reference_scale_indexes(below[:-1], distribution), kk - 1)
            and bool((edges[1:] > edges[:-1]).all())):
        raise RansError("build_indexes is not a monotone step function on this host; bin edges undefined")
    return edges


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


def device_tables(engine, distribution, z_blocks):
    """The same tables built by the GPU kernels of include/dcvc_hip.h ("update(): ... on the device", opt-in;
    SURVEY 8f-3): {"scale": (cdf, sizes, offsets), name: ...} as numpy int32 arrays shaped like the host
    builders' (columns trimmed to the longest row + 2).  z_blocks: {name: (11, C) device parameter block}.
    Device libm differs from torch-CPU's in the last ulp, so an entry on a rounding boundary may differ by one
    count from the reference table: use for streams this library both writes and reads."""
    import ctypes as C

    L = engine.L
    cols = int(L.dcvc_cdf_table_cols())
    dev = engine.device

    def fetch(rows, launch):
        cdf = torch.zeros((rows, cols), dtype=torch.int32, device=dev)
        sz = torch.zeros(rows, dtype=torch.int32, device=dev)
        off = torch.zeros(rows, dtype=torch.int32, device=dev)
        launch(cdf, sz, off)
        sz_h = sz.cpu().numpy()
        w = int(sz_h.max())
        return np.ascontiguousarray(cdf[:, :w].cpu().numpy()), sz_h, off.cpu().numpy()

    lmin, _ = scale_log_params(distribution)
    table = torch.exp(torch.linspace(lmin, math.log(SCALE_MAX), SCALE_LEVELS)).to(dev)  # the 256 levels (:220-221)
    out = {"scale": fetch(SCALE_LEVELS, lambda c, s, o: lib.check(
        L.dcvc_build_scale_cdfs(table.data_ptr(), SCALE_LEVELS, 0 if distribution == "laplace" else 1, c.data_ptr(),
                                s.data_ptr(), o.data_ptr(), engine.stream()), "build_scale_cdfs"))}
    for name, blk in z_blocks.items():
        out[name] = fetch(blk.shape[1], lambda c, s, o, blk=blk: lib.check(
            L.dcvc_build_factorized_cdfs(blk.data_ptr(), blk.shape[1], c.data_ptr(), s.data_ptr(), o.data_ptr(),
                                         engine.stream()), "build_factorized_cdfs"))
    return out


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


# ----------------------------------------------------------------------------- device coder (opt-in)
DRANS_MAGIC = b"DGR1"


class DevicePayload:
    """Payload of one picture being produced on the device: finish() waits for the size, copies
    exactly that many bytes to the host and checks the kernels' status word."""

    def __init__(self, coder, buf, size_host, event, cursor_at):
        self.coder, self.buf, self.size_host, self.event, self.cursor_at = coder, buf, size_host, event, cursor_at
        self._bytes = None
        self._streams = None  # same attribute PendingStream uses for "retired"

    def finish(self) -> bytes:
        if self._bytes is None:
            self.event.synchronize()
            words, status = int(self.size_host[self.cursor_at]), int(self.size_host[1])
            if status:
                raise RansError(f"device entropy coder status {status} (1 index, 2 space, 4 stream)")
            self._bytes = self.buf[:words].cpu().numpy().tobytes()
            self._streams = [self._bytes]
        return self._bytes

    def finish_all(self):
        return [self.finish()]


class DeviceCoder:
    """Lane-interleaved rANS on the GPU (include/dcvc_hip_rans.h): the symbol planes are coded where
    the kernels left them and only the payload crosses PCIe.  Opt-in: the wire format is NOT the
    reference's (see the header); tables are the same host-built integer CDFs, uploaded once."""

    def __init__(self, device, tables, capacity_words=8 * 1024 * 1024, symbols_per_lane=512):
        """symbols_per_lane: the size / latency knob of the format (12 bytes per lane, ~1 us per symbol
        of a lane); encoder and decoder must use the same value."""
        self.device = torch.device(device)
        self.symbols_per_lane = int(symbols_per_lane)
        self.L = lib.hip()
        self.t = {}
        for name, (cdf, ln, off) in tables.items():
            cdf, ln = np.ascontiguousarray(cdf, np.int32), np.ascontiguousarray(ln, np.int32)
            lut = np.empty((cdf.shape[0], 256), np.uint8)
            lib.check(self.L.dcvc_drans_build_lut(cdf.ctypes.data, cdf.shape[0], cdf.shape[1], ln.ctypes.data, lut.ctypes.data),
                      "drans_build_lut")
            c = torch.from_numpy(cdf).to(self.device)
            self.t[name] = (c, torch.from_numpy(ln).to(self.device),
                            torch.from_numpy(np.ascontiguousarray(off, np.int32)).to(self.device), c.shape[0], c.shape[1],
                            torch.from_numpy(lut).to(self.device))
        # two payload buffers: picture t+1 may be enqueued before picture t's bytes are fetched
        self.payloads = [torch.empty(capacity_words, dtype=torch.int32, device=self.device) for _ in range(2)]
        self.payload = self.payloads[0]
        # [cursor A, status, cursor B]: every launch reads one cursor and writes the other (its lanes are
        # spread over several workgroups); `sel` is the index of the cursor that is current
        self.state = torch.zeros(3, dtype=torch.int32, device=self.device)
        self.sel = 0
        self.state_host = [torch.zeros(3, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.init_host = torch.tensor([1, 0, 1], dtype=torch.int32).pin_memory()
        self.magic = torch.tensor([int.from_bytes(DRANS_MAGIC, "little")], dtype=torch.int32).pin_memory()
        self.flip = 0
        self._slot_owner = [None, None]   # DevicePayload that last wrote payloads[k] / state_host[k]
        self.scratch = None
        self.words = 0

    def lanes_for(self, n):
        if self.symbols_per_lane == 512:
            return int(self.L.dcvc_drans_default_lanes(n))
        lanes = (-(-n // self.symbols_per_lane) + 63) // 64 * 64
        return max(64, min(8192, lanes))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- encoder
    def _cursors(self):
        """(pointer to the current cursor, pointer to the one the next launch writes); flips."""
        a = self.state.data_ptr() + 8 * self.sel
        self.sel ^= 1
        return a, self.state.data_ptr() + 8 * self.sel

    def begin(self):
        # two payload buffers / pinned state words alternate: the picture that last used this slot must
        # have been fetched (DevicePayload.finish) before the slot is rewritten, or its bytes are lost
        owner = self._slot_owner[self.flip]
        if owner is not None and owner._bytes is None:
            raise RuntimeError("more than two device-coded pictures are pending: finish() the oldest first")
        self.payload = self.payloads[self.flip]
        self.payload[:1].copy_(self.magic, non_blocking=True)
        self.state.copy_(self.init_host, non_blocking=True)
        self.sel = 0

    def encode(self, table, sym: torch.Tensor, idx=None, chan=None, lanes=None):
        cdf, ln, off, rows, stride, _ = self.t[table]
        n = sym.numel()
        lanes = lanes or self.lanes_for(n)
        need = int(self.L.dcvc_drans_scratch_words(n, lanes))
        if self.scratch is None or self.scratch.numel() < need:
            self.scratch = torch.empty(need, dtype=torch.int32, device=self.device)
        chw, cc = (0, 0) if chan is None else (chan[1], chan[0])
        lib.check(self.L.dcvc_drans_encode(sym.data_ptr(), idx.data_ptr() if idx is not None else None, chw, cc, n,
                                           cdf.data_ptr(), rows, stride, ln.data_ptr(), off.data_ptr(), lanes,
                                           self.scratch.data_ptr(), self.scratch.numel(), self.payload.data_ptr(),
                                           self.payload.numel(), *self._cursors(), self.state.data_ptr() + 4,
                                           self._stream()), "drans_encode")

    def end(self) -> DevicePayload:
        host = self.state_host[self.flip]
        self.flip ^= 1
        host.copy_(self.state, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        p = DevicePayload(self, self.payload, host, ev, 2 * self.sel)
        self._slot_owner[self.flip ^ 1] = p
        return p

    # -- decoder
    DEC_SLOTS = 4

    def set_stream(self, data: bytes):
        """Upload a payload through a pinned staging buffer on a copy stream, so a decoder that defers
        its status check (decompress(..., defer_check=True)) never blocks the host on the GPU."""
        if len(data) < 4 or data[:4] != DRANS_MAGIC or len(data) % 4:
            raise RansError("not a device-coder payload (magic DGR1 missing)")
        words = len(data) // 4
        if not hasattr(self, "dec"):
            self.dec = [dict(dev=None, pin=None, h2d=None, free=None) for _ in range(self.DEC_SLOTS)]
            self.dec_flip = 0
            self.copy_stream = torch.cuda.Stream(self.device)
            self.one_host = torch.tensor([1], dtype=torch.int32).pin_memory()
        k = self.dec_flip
        self.dec_flip = (k + 1) % self.DEC_SLOTS
        d = self.dec[k]
        if d["h2d"] is not None:
            d["h2d"].synchronize()          # the staging buffer's previous upload has left the host
        if d["pin"] is None or d["pin"].numel() < words:
            cap = max(words, 1 << 20)
            d["pin"] = torch.empty(cap, dtype=torch.int32).pin_memory()
            # allocated ON the copy stream: a block the caching allocator recycles from the launch
            # stream may still be written by kernels the host has run ahead of, and the copy stream
            # does not wait for those
            with torch.cuda.stream(self.copy_stream):
                d["dev"] = torch.empty(cap, dtype=torch.int32, device=self.device)
        d["pin"][:words].copy_(torch.frombuffer(bytearray(data), dtype=torch.int32))
        main = torch.cuda.current_stream(self.device)
        if d["free"] is not None:
            self.copy_stream.wait_event(d["free"])   # kernels that read this slot's previous payload are done
        with torch.cuda.stream(self.copy_stream):
            d["dev"][:words].copy_(d["pin"][:words], non_blocking=True)
            d["h2d"] = torch.cuda.Event()
            d["h2d"].record(self.copy_stream)
        main.wait_event(d["h2d"])
        self.payload, self.words, self.cur = d["dev"], words, d
        self.state[:1].copy_(self.one_host, non_blocking=True)   # cursor behind the magic; status stays sticky
        self.sel = 0

    def release(self):
        """Call after the last decode() of a payload: its slot may be overwritten once they have run."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.cur["free"] = ev

    def decode(self, table, n, idx=None, chan=None, lanes=None) -> torch.Tensor:
        cdf, ln, off, rows, stride, lut = self.t[table]
        lanes = lanes or self.lanes_for(n)
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        chw, cc = (0, 0) if chan is None else (chan[1], chan[0])
        lib.check(self.L.dcvc_drans_decode(self.payload.data_ptr(), self.words, *self._cursors(),
                                           idx.data_ptr() if idx is not None else None, chw, cc, n, cdf.data_ptr(), rows,
                                           stride, ln.data_ptr(), off.data_ptr(), lut.data_ptr(), lanes, out.data_ptr(),
                                           self.state.data_ptr() + 4, self._stream()), "drans_decode")
        return out

    def check(self):
        """Synchronising: raise if any decode launch since the last check() flagged bad input."""
        status = int(self.state[1].item())
        if status:
            self.state[1:2].zero_()
            raise RansError(f"device entropy coder status {status} (1 index, 2 space, 4 stream)")
