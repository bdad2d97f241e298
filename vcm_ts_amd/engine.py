"""Device-side plumbing: strided-NHWC buffer views, a named workspace, packed-weight cache
and thin Python wrappers over the C ABI of include/dcvc_hip.h.

PyTorch is used for exactly three things here: allocating HBM (torch.empty on the GPU),
naming the HIP stream kernels are enqueued on (torch.cuda.current_stream) and exposing
results as tensors.  All arithmetic on the product path happens in libdcvc_hip.so; there is
no eager/CPU fallback -- without the library or without a GPU the constructors raise.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import lib

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def _r4(c):
    return (c + 3) // 4 * 4


class View:
    """C channels starting at channel `coff` of an fp32 (N, H, W, cs) buffer."""

    __slots__ = ("base", "N", "H", "W", "C", "cs", "coff", "ptr")

    def __init__(self, base: torch.Tensor, C_: int, coff: int = 0, geom=None):
        self.base = base
        if geom is None:
            assert base.dim() == 4 and base.dtype == torch.float32 and base.is_contiguous()
            self.N, self.H, self.W, self.cs = base.shape
            self.ptr = base.data_ptr() + 4 * coff
        else:  # foreign channels-last tensor: (N, H, W, cs, ptr); `base` only keeps it alive
            self.N, self.H, self.W, self.cs, self.ptr = geom
        self.C = C_
        self.coff = coff
        assert coff + C_ <= self.cs

    @staticmethod
    def alias(t: torch.Tensor):
        """View over a logical-NCHW tensor whose memory already is strided NHWC (one of our own
        outputs handed back by the caller), or None if its layout does not qualify."""
        if t.dim() != 4 or t.dtype != torch.float32 or not t.is_cuda:
            return None
        N, C_, H, W = t.shape
        cs = t.stride(3)
        if t.stride(1) == 1 and cs >= C_ and cs % 4 == 0 and t.stride(2) == W * cs and t.stride(0) == H * W * cs \
                and t.data_ptr() % 16 == 0:
            return View(t, C_, 0, geom=(N, H, W, cs, t.data_ptr()))
        return None

    def slice(self, c0: int, c: int) -> "View":
        v = View(self.base, c, 0, geom=(self.N, self.H, self.W, self.cs, self.ptr + 4 * c0))
        v.coff = self.coff + c0
        return v

    def nchw(self) -> torch.Tensor:
        """Zero-copy logical (N, C, H, W) tensor over this view (channels-last strides)."""
        if self.base.dim() == 4 and tuple(self.base.shape) == (self.N, self.H, self.W, self.cs):
            return self.base[..., self.coff : self.coff + self.C].permute(0, 3, 1, 2)
        return self.base[:, self.coff : self.coff + self.C]  # alias of a caller's logical-NCHW tensor

    @property
    def HW(self):
        return self.H * self.W

    def __repr__(self):
        return f"View(N={self.N},H={self.H},W={self.W},C={self.C},cs={self.cs},coff={self.coff})"


class PackedConv:
    __slots__ = ("w", "b", "ks", "Cout", "Cout_pad", "seg_C", "ps", "version", "precision", "weight", "bias",
                 "cin_slice", "key", "small", "k32", "paired", "host", "job")


class Engine:
    def __init__(self, device, precision=None):
        import os

        self.precision = precision or os.environ.get("DCVC_PRECISION", "fp32")
        if self.precision not in lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(lib.PRECISIONS)}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP engine needs a GPU device (no CPU fallback exists)")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.L = lib.hip()
        self.bufs = {}
        self.packs = {}
        self._dev_packs = []   # device-packed layers in creation order (pack_dev): the jobs of repack_all()'s plan
        self._plan = None      # (handle, number of jobs, weight pointers) of dcvc_pack_plan_*
        self.calls = 0
        self.profile = None  # set to {} to time every conv launch with HIP events (bench.py)
        self.profile_detail = None
        self.profile_hbm = None  # set to {} to time the HBM-bound kernels (warp, resamplers, dual prior, layout) too
        self.tape = None     # grad.Tape while a training-mode forward is being recorded
        self._edges = {}     # distribution -> device (256,) fp32 bin edges of build_indexes
        self._status = None  # device status word of the split-fp16 kernels (saturation flag)
        # Which kernel serves a layer (conv_mfma / conv_k32 / conv_small / the tap-paired 7x7) is part of the arithmetic
        # -- each sums in its own order -- so an encoder and its decoder must route alike.  The product routes by layer
        # geometry alone; these switches exist for developer A/B runs (tools/) and for tests, and are read from the
        # environment only when DCVC_DEV=1 says the process is one of those (a stray variable must not be
        # able to make two product processes disagree).
        dev = os.environ.get("DCVC_DEV") == "1"
        sw = lambda name, default: os.environ.get(name, default) if dev else default
        self.use_small = sw("DCVC_SMALL", "1") != "0"   # dcvc_conv2d_small for <= 16 output channels
        self.use_k32 = sw("DCVC_K32", "1") != "0"       # dcvc_conv2d_k32 (16x16x32 MFMA, 32-channel chunks)
        self.use_pairs = sw("DCVC_PAIR_TAPS", "1") != "0"  # tap-paired 7x7 kernel for layers with <= 8 input channels
        self.k32_everywhere = False  # tests: route every layer the kernel covers to it, whatever its size
        self.k32_sizes = tuple(int(k) for k in sw("DCVC_K32_SIZES", "3").split(","))  # kernel sizes it takes (1x1 layers are HBM-bound: conv_mfma's full-line stores are 10-15 % faster there)
        # fp16x3 mode clamps |activation| > 8188 on load; every convolution launch of that mode flags outputs beyond
        # that magnitude in the status word (read_status() / check_status()).  The check is a running maximum (two
        # v_max3_f32 per four outputs, one atomic per workgroup only when it fires) and is ON by default since round 3;
        # DCVC_RANGE_CHECK=0 switches it off for the kernels where it is optional (conv_mfma, conv_small).
        self.range_check = os.environ.get("DCVC_RANGE_CHECK", "1") == "1"
        self.guard_outputs = True  # grad.Tape.backward clears it around its data-gradient launches
        # fast mode: weight gradients of stride-1 layers on the bf16 matrix cores (hi + lo operands, three products);
        # DCVC_WGRAD_SPLIT=0 keeps them on the fp32 MFMA
        self.wgrad_split = os.environ.get("DCVC_WGRAD_SPLIT", "1") != "0"
        self.wgrad_side_stream = sw("DCVC_WGRAD_SIDE", "1") != "0"  # weight gradients on a second stream beside the data-gradient chain

    # ------------------------------------------------------------------ memory
    def stream(self):
        # the raw handle of torch's current stream on this device (torch.cuda.current_stream() builds a Stream object
        # per call: ~5 us, a few milliseconds per training step at ~800 launches)
        return C.c_void_p(_raw_stream(self._dev_index))

    def _store(self, scratch=False):
        """Inference recycles one named workspace; a recorded (training) forward owns fresh buffers
        that live as long as its tape, because backward reads every intermediate."""
        return self.bufs if (self.tape is None or scratch) else self.tape.arena

    def buf(self, name, N, H, W, C_, cs=None, zero=False) -> View:
        cs = cs or _r4(C_)
        key = (name, N, H, W, cs)
        store = self._store()
        t = store.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)((N, H, W, cs), dtype=torch.float32, device=self.device)
            store[key] = t
        return View(t, C_)

    def status_word(self) -> torch.Tensor:
        if self._status is None:
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._status

    def read_status(self) -> int:
        """Status word of the split-fp16 kernels since the last read (synchronises); 0 = nothing clamped."""
        if self._status is None:
            return 0
        v = int(self._status.item())
        if v:
            self._status.zero_()
        return v

    def check_status(self):
        """Raise if a split-fp16 kernel clamped (or would clamp) an activation since the last call."""
        v = self.read_status()
        if v:
            raise lib.KernelError(f"split-fp16 activation range exceeded (status {v}): |x| > 8188; use precision='fp32'")

    def status_snapshot(self):
        """Asynchronous form of check_status() for pipelined callers (GopEncoder): copies the status word to pinned host
        memory behind everything enqueued so far on the current stream and returns a callable that waits for THAT copy
        only (not for later work) and raises like check_status().  None when no split-fp16 kernel has run."""
        if self._status is None:
            return None
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(self._status, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))

        def check():
            ev.synchronize()
            v = int(host[0])
            if v:
                self._status.zero_()
                raise lib.KernelError(f"split-fp16 activation range exceeded (status {v}): |x| > 8188; use precision='fp32'")

        return check

    def ibuf(self, name, n) -> torch.Tensor:
        key = (name, n, "i32")
        store = self._store()
        t = store.get(key)
        if t is None:
            t = torch.empty(n, dtype=torch.int32, device=self.device)
            store[key] = t
        return t

    def fbuf(self, name, n, scratch=False) -> torch.Tensor:
        key = (name, n, "f32")
        store = self._store(scratch)
        t = store.get(key)
        if t is None:
            t = torch.empty(n, dtype=torch.float32, device=self.device)
            store[key] = t
        return t

    def _rec(self, *op):
        if self.tape is not None:
            self.tape.ops.append(op)

    def _rec_unary(self, name, src, out, *rest):
        """Resamplings of a gradient-free input (picture pyramids) stay gradient-free."""
        if self.tape is not None:
            if self.tape.is_const(src):
                self.tape.mark_const(out)
            else:
                self.tape.ops.append((name, src, out) + rest)

    def release(self):
        self.bufs.clear()

    def __del__(self):  # the pack plan owns a small device table (dcvc_pack_plan_create)
        plan, self._plan = getattr(self, "_plan", None), None
        if plan is not None:
            try:
                self.L.dcvc_pack_plan_destroy(plan[0])
            except Exception:
                pass

    def bytes_reserved(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())

    # ------------------------------------------------------------------ boundary layout
    def from_nchw(self, x: torch.Tensor, out: View) -> View:
        """NCHW (any strides) -> out view.  A tensor that already is one of our channels-last
        views is still copied: callers own their tensors and may mutate them."""
        N, C_, H, W = x.shape
        assert (out.N, out.H, out.W, out.C) == (N, H, W, C_), (x.shape, out)
        x = x.detach()
        if x.device != self.device or x.dtype != torch.float32:
            x = x.to(device=self.device, dtype=torch.float32)
        cs = x.stride(3) if x.dim() == 4 else 0
        if x.stride(1) == 1 and cs >= C_ and cs % 4 == 0 and x.stride(2) == W * cs and x.stride(0) == H * W * cs \
                and x.data_ptr() % 16 == 0:
            lib.check(self.L.dcvc_copy_channels(x.data_ptr(), cs, out.ptr, out.cs, N * H * W, C_, self.stream()),
                      "copy_channels")
        else:
            x = x.contiguous()
            self._hbm("nchw_to_nhwc", f"C{C_} {H}x{W}", N * C_ * H * W * 8,
                      lambda: lib.check(self.L.dcvc_nchw_to_nhwc(x.data_ptr(), out.ptr, out.cs, N, C_, H, W, self.stream()),
                                        "nchw_to_nhwc"))
        self.calls += 1
        return out

    def to_nchw(self, v: View, clamp01=False) -> torch.Tensor:
        out = torch.empty((v.N, v.C, v.H, v.W), dtype=torch.float32, device=self.device)
        lib.check(self.L.dcvc_nhwc_to_nchw(v.ptr, v.cs, out.data_ptr(), v.N, v.C, v.H, v.W, int(clamp01),
                                           self.stream()), "nhwc_to_nchw")
        return out

    # ------------------------------------------------------------------ convolution
    def pack_dev(self, key, weight: torch.Tensor, bias, seg_C, ps, cin_slice=None, transposed=False) -> PackedConv:
        """Device-side packing (dcvc_conv_pack_weights_dev): used while training, where the weights
        change every optimiser step, and for the transposed filters of the data gradient."""
        ver = (weight._version, None if bias is None else bias._version, weight.data_ptr())
        key = (key, self.precision, "dev", bool(transposed))
        pk = self.packs.get(key)
        if pk is not None and pk.version == ver:
            return pk
        Cout, CinT, ks, _ = weight.shape
        off = 0 if cin_slice is None else cin_slice[0]
        if transposed:
            assert len(seg_C) == 1 and not ps
            pk_cout, pk_segs = seg_C[0], (Cout,)
        else:
            pk_cout, pk_segs = Cout, tuple(seg_C)
        if pk is None:
            segs = (C.c_int32 * len(pk_segs))(*pk_segs)
            cpad = C.c_int32()
            n = self.L.dcvc_conv_pack_size(pk_cout, ks, len(pk_segs), segs, C.byref(cpad))
            if n < 0:
                raise lib.KernelError(f"conv_pack_size({key})")
            pk = PackedConv()
            pk.w = torch.empty(n, dtype=torch.float32, device=self.device)
            pk.b = torch.empty(cpad.value, dtype=torch.float32, device=self.device)
            pk.ks, pk.Cout, pk.Cout_pad, pk.seg_C, pk.ps = ks, pk_cout, cpad.value, pk_segs, bool(ps)
            pk.precision = lib.PRECISIONS[self.precision]
            pk.key = key
            self.packs[key] = pk
            self._dev_packs.append(pk)
        assert weight.is_contiguous() and weight.dtype == torch.float32 and weight.device == self.device
        segs = (C.c_int32 * len(seg_C))(*seg_C)
        lib.check(self.L.dcvc_conv_pack_weights_dev(weight.data_ptr(), None if (bias is None or transposed) else bias.data_ptr(),
                                                    Cout, CinT, ks, len(seg_C), segs, off, int(ps), pk.precision,
                                                    int(transposed), pk.w.data_ptr(), pk.b.data_ptr(), self.stream()),
                  f"conv_pack_weights_dev({key})")
        pk.version = ver
        pk.weight, pk.bias, pk.cin_slice = weight, bias, cin_slice
        pk.job = (weight, None if (bias is None or transposed) else bias, Cout, CinT, ks, tuple(seg_C), off, int(ps),
                  int(transposed))
        return pk

    def repack_all(self):
        """Training: bring every device-packed filter (forward and data-gradient packings) up to date with ONE launch.
        The optimiser changes every weight every step, and ~370 per-layer packing launches cost more host time than
        anything else in a batch-4 256x256 step; the plan (dcvc_pack_plan_*) holds all their jobs on the device.
        Called at the start of a recorded forward; layers first seen later in the step pack themselves as before
        (pack_dev) and join the plan at the next call."""
        packs = self._dev_packs
        if len(packs) < 8:
            return
        vers, stale = [], False
        for pk in packs:
            w, b = pk.job[0], pk.job[1]
            ver = (w._version, None if pk.bias is None else pk.bias._version, w.data_ptr())
            stale = stale or pk.version != ver
            vers.append(ver)
        if not stale:
            return
        ptrs = tuple(v[2] for v in vers)
        if self._plan is None or self._plan[1] != len(packs) or self._plan[2] != ptrs:
            if self._plan is not None:
                self.L.dcvc_pack_plan_destroy(self._plan[0])
                self._plan = None
            jobs = (lib.PackJob * len(packs))()
            for j, pk in zip(jobs, packs):
                w, b, Cout, CinT, ks, seg_C, off, ps, transposed = pk.job
                j.w, j.b = w.data_ptr(), None if b is None else b.data_ptr()
                j.Cout, j.Cin_total, j.ks, j.nseg = Cout, CinT, ks, len(seg_C)
                for i, c in enumerate(seg_C):
                    j.seg_C[i] = c
                j.cin_offset, j.pixel_shuffle, j.precision, j.transposed = off, ps, pk.precision, transposed
                j.wpack, j.bpack = pk.w.data_ptr(), pk.b.data_ptr()
            handle = C.c_void_p()
            torch.cuda.synchronize(self.device)  # create copies the table with a blocking call
            lib.check(self.L.dcvc_pack_plan_create(jobs, len(packs), C.byref(handle)), "pack_plan_create")
            self._plan = (handle, len(packs), ptrs)
        lib.check(self.L.dcvc_pack_plan_run(self._plan[0], self.stream()), "pack_plan_run")
        for pk, ver in zip(packs, vers):
            pk.version = ver

    def pack(self, key, weight: torch.Tensor, bias, seg_C, ps, cin_slice=None) -> PackedConv:
        if self.tape is not None:
            return self.pack_dev(key, weight, bias, seg_C, ps, cin_slice)
        ver = (weight._version, None if bias is None else bias._version, weight.data_ptr())
        key = (key, self.precision)
        pk = self.packs.get(key)
        if pk is not None and pk.version == ver:
            return pk
        w = weight.detach().float().cpu()
        if cin_slice is not None:
            w = w[:, cin_slice[0] : cin_slice[1]]
        w = w.contiguous().numpy()
        Cout, Cin, ks, _ = w.shape
        assert sum(seg_C) == Cin, (key, seg_C, Cin)
        b = None if bias is None else bias.detach().float().cpu().contiguous().numpy()
        segs = (C.c_int32 * len(seg_C))(*seg_C)
        cpad = C.c_int32()
        n = self.L.dcvc_conv_pack_size(Cout, ks, len(seg_C), segs, C.byref(cpad))
        if n < 0:
            raise lib.KernelError(f"conv_pack_size({key})")
        wp = np.empty(n, np.float32)
        bp = np.empty(cpad.value, np.float32)
        lib.check(self.L.dcvc_conv_pack_weights(w.ctypes.data, None if b is None else b.ctypes.data, Cout, ks,
                                                len(seg_C), segs, int(ps), lib.PRECISIONS[self.precision],
                                                wp.ctypes.data, bp.ctypes.data),
                  f"conv_pack_weights({key})")
        pk = PackedConv()
        pk.w = torch.from_numpy(wp).to(self.device)
        pk.b = torch.from_numpy(bp).to(self.device)
        pk.ks, pk.Cout, pk.Cout_pad, pk.seg_C, pk.ps, pk.version = ks, Cout, cpad.value, tuple(seg_C), bool(ps), ver
        pk.precision = lib.PRECISIONS[self.precision]
        pk.weight, pk.bias, pk.cin_slice, pk.key = weight, bias, cin_slice, key
        pk.host = True  # packed from the module's own weight: the alternative kernels may re-pack it their way
        self.packs[key] = pk
        return pk

    def pack_small(self, pk: PackedConv) -> PackedConv:
        """The same layer packed for dcvc_conv2d_small (<= 16 output channels), cached beside the other packing."""
        q = getattr(pk, "small", None)
        if q is not None and q.version == pk.version:
            return q
        w = pk.weight.detach().float().cpu()
        if pk.cin_slice is not None:
            w = w[:, pk.cin_slice[0] : pk.cin_slice[1]]
        w = w.contiguous().numpy()
        b = None if pk.bias is None else pk.bias.detach().float().cpu().contiguous().numpy()
        segs = (C.c_int32 * len(pk.seg_C))(*pk.seg_C)
        n = self.L.dcvc_conv_small_pack_bytes(pk.Cout, pk.ks, len(pk.seg_C), segs)
        if n < 0:
            raise lib.KernelError(f"conv_small_pack_bytes({pk.key})")
        wp = np.empty(n // 4, np.float32)
        bp = np.empty(16, np.float32)
        lib.check(self.L.dcvc_conv_small_pack_weights(w.ctypes.data, None if b is None else b.ctypes.data, pk.Cout, pk.ks,
                                                      len(pk.seg_C), segs, wp.ctypes.data, bp.ctypes.data),
                  f"conv_small_pack_weights({pk.key}) [status -3: a |weight| >= 1023.5 does not fit split fp16, use precision='fp32']")
        q = PackedConv()
        q.w, q.b = torch.from_numpy(wp).to(self.device), torch.from_numpy(bp).to(self.device)
        q.ks, q.Cout, q.Cout_pad, q.seg_C, q.ps, q.version, q.key = pk.ks, pk.Cout, 16, pk.seg_C, False, pk.version, pk.key
        pk.small = q
        return q

    def pack_paired(self, pk: PackedConv) -> PackedConv:
        """The same 7x7 layer (one segment of <= 8 input channels) packed in tap pairs for dcvc_conv2d with pair_taps."""
        q = getattr(pk, "paired", None)
        if q is not None and q.version == pk.version:
            return q
        w = pk.weight.detach().float().cpu()
        if pk.cin_slice is not None:
            w = w[:, pk.cin_slice[0] : pk.cin_slice[1]]
        w = w.contiguous().numpy()
        b = None if pk.bias is None else pk.bias.detach().float().cpu().contiguous().numpy()
        cpad = C.c_int32()
        n = self.L.dcvc_conv_pack_size_paired(pk.Cout, pk.seg_C[0], C.byref(cpad))
        if n < 0:
            raise lib.KernelError(f"conv_pack_size_paired({pk.key})")
        wp = np.empty(n, np.float32)
        bp = np.empty(cpad.value, np.float32)
        lib.check(self.L.dcvc_conv_pack_weights_paired(w.ctypes.data, None if b is None else b.ctypes.data, pk.Cout, pk.seg_C[0],
                                                       wp.ctypes.data, bp.ctypes.data),
                  f"conv_pack_weights_paired({pk.key}) [status -3: a |weight| >= 1023.5 does not fit split fp16, use precision='fp32']")
        q = PackedConv()
        q.w, q.b = torch.from_numpy(wp).to(self.device), torch.from_numpy(bp).to(self.device)
        q.ks, q.Cout, q.Cout_pad, q.seg_C, q.ps, q.version, q.key = pk.ks, pk.Cout, cpad.value, pk.seg_C, False, pk.version, pk.key
        pk.paired = q
        return q

    def pair_capable(self, pk: PackedConv, stride) -> bool:
        """7x7 stride-1 layers with ONE input segment of <= 8 channels (SpyNet's first layer, MEBasic conv1) in fast mode:
        two taps share a 16-deep K step (dcvc_conv_args.pair_taps).  Geometry of the layer only: the same decision on the
        encoder and the decoder side.  Host-packed weights only (a training step packs on the device, plain layout)."""
        return (self.precision == "fp16x3" and self.tape is None and self.use_pairs and getattr(pk, "host", False)
                and pk.ks == 7 and stride == 1 and len(pk.seg_C) == 1 and pk.seg_C[0] <= 8 and not pk.ps and pk.Cout > 16)

    def pack_k32(self, pk: PackedConv) -> PackedConv:
        """The same layer packed for dcvc_conv2d_k32 (32-channel chunks), cached beside the other packing."""
        q = getattr(pk, "k32", None)
        if q is not None and q.version == pk.version:
            return q
        w = pk.weight.detach().float().cpu()
        if pk.cin_slice is not None:
            w = w[:, pk.cin_slice[0] : pk.cin_slice[1]]
        w = w.contiguous().numpy()
        b = None if pk.bias is None else pk.bias.detach().float().cpu().contiguous().numpy()
        segs = (C.c_int32 * len(pk.seg_C))(*pk.seg_C)
        cpad = C.c_int32()
        n = self.L.dcvc_conv_k32_pack_bytes(pk.Cout, pk.ks, len(pk.seg_C), segs, C.byref(cpad))
        if n < 0:
            raise lib.KernelError(f"conv_k32_pack_bytes({pk.key})")
        wp = np.empty(n // 4, np.float32)
        bp = np.empty(cpad.value, np.float32)
        lib.check(self.L.dcvc_conv_k32_pack_weights(w.ctypes.data, None if b is None else b.ctypes.data, pk.Cout, pk.ks,
                                                    len(pk.seg_C), segs, int(pk.ps), wp.ctypes.data, bp.ctypes.data),
                  f"conv_k32_pack_weights({pk.key}) [status -3: a |weight| >= 1023.5 does not fit split fp16, use precision='fp32']")
        q = PackedConv()
        q.w, q.b = torch.from_numpy(wp).to(self.device), torch.from_numpy(bp).to(self.device)
        q.ks, q.Cout, q.Cout_pad, q.seg_C, q.ps, q.version, q.key = pk.ks, pk.Cout, cpad.value, pk.seg_C, pk.ps, pk.version, pk.key
        pk.k32 = q
        return q

    def k32_capable(self, pk: PackedConv, stride, out: View, res, res2, gate, srcs=()) -> bool:
        """Layers routed to dcvc_conv2d_k32: fp16x3, 3x3, stride 1, every input segment a multiple of 32 channels,
        16-byte-addressable epilogue (4-channel groups of out / residuals) -- what the kernel covers -- and, from the
        per-layer comparison inside a 1080p P picture (profiles/r03_k32_vs_conv_mfma_in_pipeline.txt), only where it
        wins: output-channel blocks of 64 (its 32-column variant is 25-40 % slower than conv_mfma's) and at least
        ~1000 workgroups (below that the two are within noise or conv_mfma is ahead).  Geometry only: an encoder and
        its decoder take the same decision."""
        if not (self.precision == "fp16x3" and self.tape is None and self.use_k32 and getattr(pk, "host", False)):
            return False
        cfin = pk.Cout // 4 if pk.ps else pk.Cout
        al = lambda v: v is None or (v.ptr % 16 == 0 and v.cs % 4 == 0)
        covered = (pk.ks in self.k32_sizes and stride == 1 and all(c % 32 == 0 for c in pk.seg_C) and cfin % 4 == 0
                   and al(out) and al(res) and al(res2) and (gate is None or gate.data_ptr() % 16 == 0)
                   # what dcvc_conv2d_k32 itself refuses: misaligned source views, and pictures whose
                   # pixel index or channel stride does not fit its 24-bit address multiplies -- those layers stay
                   # on dcvc_conv2d instead of failing with DCVC_E_ARG
                   and all(al(s) for s in srcs) and out.H * out.W < (1 << 24)
                   and all(v is None or v.cs * 4 < (1 << 24) for v in (out, res, res2, *srcs)))
        if not covered or self.k32_everywhere:
            return covered
        # per-IMAGE geometry, never the batch size: a batch of rate points is decoded one element at a time
        # (test_config_c5_four_rate_points_at_bench_size caught a batch-dependent rule)
        m = 2 if pk.ps else 1
        tiles = ((out.H // m + 7) // 8) * ((out.W // m + 31) // 32)
        return pk.Cout_pad % 64 == 0 and tiles * (pk.Cout_pad // 64) >= 1000

    def small_capable(self, pk: PackedConv, stride, gate, res2, chan_partial) -> bool:
        """Layers dcvc_conv2d_small covers: <= 16 output channels, 3x3 / 7x7, stride 1, plain epilogue, fp16x3."""
        return (self.precision == "fp16x3" and self.tape is None and self.use_small and getattr(pk, "host", False)
                and pk.Cout <= 16 and pk.ks in (3, 7)
                and stride == 1 and not pk.ps and gate is None and res2 is None and chan_partial is None)

    def conv(self, pk: PackedConv, srcs, out: View, stride=1, in_slope=None, out_slope=None, res: View = None,
             gate: torch.Tensor = None, res2: View = None, chan_partial: torch.Tensor = None, band=None):
        """chan_partial: buffer from chan_partial_buf() that receives the per-workgroup channel sums of the output
        (fused SE squeeze).  Which kernel serves the layer (conv_mfma / conv_k32 / conv_small) is decided in
        _conv_f32 from the layer's geometry alone, so an encoder and its decoder always agree."""
        return self._conv_f32(pk, srcs, out, stride, in_slope, out_slope, res, gate, res2, chan_partial, band)

    def _launch_conv(self, launch, pk, s0, Ho, Wo, stride, res, res2, tag="", note=""):
        if self.profile is None:
            launch()
            return
        # events go on the stream the kernel is launched on (torch's current stream)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        launch()
        ev1.record()
        flops = 2.0 * s0.N * Ho * Wo * pk.Cout * sum(pk.seg_C) * pk.ks * pk.ks  # algorithmic, unpadded
        nout = s0.N * Ho * Wo * pk.Cout
        abytes = 4.0 * (s0.N * s0.H * s0.W * sum(pk.seg_C) + nout * (1 + (res is not None) + (res2 is not None))
                        + pk.Cout * sum(pk.seg_C) * pk.ks * pk.ks)  # each operand once
        self.profile.setdefault(f"conv{pk.ks}x{pk.ks}s{stride}{tag}", []).append((ev0, ev1, flops, abytes))
        if self.profile_detail is not None:
            self.profile_detail.append((ev0, ev1, flops, f"k{pk.ks}s{stride}{tag} {pk.seg_C}->{pk.Cout}{'ps' if pk.ps else ''} "
                                                         f"{s0.H}x{s0.W}", note))

    def chan_partial_buf(self, name, pk: PackedConv, out: View, stride=1):
        """(buffer, rows per image) for the fused channel sums of a convolution writing `out`."""
        parts = int(self.L.dcvc_conv_chan_partial_parts(pk.ks, stride, out.H, out.W))
        if parts <= 0:
            raise lib.KernelError("conv_chan_partial_parts")
        return self.fbuf(name + ".chan_partial", out.N * parts * pk.Cout_pad), parts

    def _conv_f32(self, pk: PackedConv, srcs, out: View, stride=1, in_slope=None, out_slope=None, res: View = None,
                  gate: torch.Tensor = None, res2: View = None, chan_partial: torch.Tensor = None, band=None):
        """band = (first tile row, tile rows): compute only that band of output rows (dcvc_conv_args.tile_row0 /
        tile_rows; a tile row is dcvc_conv_tile_rows(ks, stride) output rows)."""
        a = lib.ConvArgs()
        assert len(srcs) == len(pk.seg_C)
        s0 = srcs[0]
        for i, (s, c) in enumerate(zip(srcs, pk.seg_C)):
            assert s.C == c and (s.N, s.H, s.W) == (s0.N, s0.H, s0.W), (s, c)
            a.seg[i].ptr, a.seg[i].C, a.seg[i].cs = s.ptr, s.C, s.cs
        a.nseg, a.N, a.Hin, a.Win = len(srcs), s0.N, s0.H, s0.W
        a.in_act, a.in_slope = (0, 0.0) if in_slope is None else (1, in_slope)
        small = band is None and self.small_capable(pk, stride, gate, res2, chan_partial)
        k32 = not small and not isinstance(out_slope, tuple) and self.k32_capable(pk, stride, out, res, res2, gate, srcs)
        paired = not small and not k32 and self.pair_capable(pk, stride)
        wq = self.pack_small(pk) if small else (self.pack_k32(pk) if k32 else (self.pack_paired(pk) if paired else pk))
        a.pair_taps = int(paired)
        a.wpack, a.bpack = wq.w.data_ptr(), wq.b.data_ptr()
        a.ks, a.stride, a.Cout, a.Cout_pad = pk.ks, stride, pk.Cout, wq.Cout_pad
        pad = pk.ks // 2
        Ho = (s0.H + 2 * pad - pk.ks) // stride + 1
        Wo = (s0.W + 2 * pad - pk.ks) // stride + 1
        m = 2 if pk.ps else 1
        cfin = pk.Cout // 4 if pk.ps else pk.Cout
        assert (out.N, out.H, out.W, out.C) == (s0.N, Ho * m, Wo * m, cfin), (out, Ho, Wo, cfin)
        a.out, a.out_cs = out.ptr, out.cs
        if isinstance(out_slope, tuple):  # ("mask", slope): res2 is the mask source (dcvc_conv_args.out_act 3)
            assert out_slope[0] == "mask" and res2 is not None and not (small or k32)
            a.out_act, a.out_slope = 3, float(out_slope[1])
        else:
            a.out_act, a.out_slope = (0, 0.0) if out_slope is None else ((2, 0.0) if out_slope == "clamp01" else (1, out_slope))
        a.pixel_shuffle = int(pk.ps)
        a.precision = pk.precision
        if res is not None:
            assert (res.N, res.H, res.W, res.C) == (out.N, out.H, out.W, out.C)
            a.res, a.res_cs = res.ptr, res.cs
        if gate is not None:
            a.res_gate = gate.data_ptr()
        if res2 is not None:
            assert (res2.N, res2.H, res2.W, res2.C) == (out.N, out.H, out.W, out.C)
            a.res2, a.res2_cs = res2.ptr, res2.cs
        # range guard of the split-fp16 mode: inference launches only (the data-gradient launches of a training
        # backward reuse this kernel on gradients, whose magnitudes say nothing about activation range)
        if (self.range_check or k32) and self.precision == "fp16x3" and self.guard_outputs:
            a.status = self.status_word().data_ptr()
        if chan_partial is not None:
            a.chan_partial = chan_partial.data_ptr()
        if band is not None:
            a.tile_row0, a.tile_rows = int(band[0]), int(band[1])
        fn, what = (self.L.dcvc_conv2d_small, "conv2d_small") if small else (
            (self.L.dcvc_conv2d_k32, "conv2d_k32") if k32 else (self.L.dcvc_conv2d, "conv2d"))
        note = ""
        if self.profile_detail is not None:  # per-launch listing of tools/in_pipeline_detail.py: layer name and epilogue flags
            note = (f"{pk.key[0][1] if isinstance(pk.key[0], tuple) and len(pk.key[0]) > 1 else pk.key[0]}"
                    f" in_act={in_slope} out_act={out_slope} in_cs={[s.cs for s in srcs]} out_cs={out.cs}"
                    f" res={None if res is None else res.cs} gate={gate is not None} res2={None if res2 is None else res2.cs}"
                    f" chan_sums={chan_partial is not None}")
        self._launch_conv(lambda: lib.check(fn(C.byref(a), self.stream()), what), pk, s0, Ho, Wo, stride, res, res2,
                          "small" if small else ("k32" if k32 else ""), note)
        self.calls += 1
        self._rec("conv", pk, tuple(srcs), out, stride, in_slope, out_slope, res, gate, res2)
        return out

    def _hbm(self, kernel, shape, abytes, launch):
        """Launch one of the HBM-bound kernels; while profiling (bench.py sets profile_hbm = {}) bracket it with HIP
        events on the launch stream and record its ALGORITHMIC bytes (every operand once: SURVEY 8d)."""
        if self.profile_hbm is None:
            return launch()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        launch()
        ev1.record()
        self.profile_hbm.setdefault((kernel, shape), []).append((ev0, ev1, float(abytes)))

    def collect_profile_hbm(self):
        """[{kernel, shape, launches, avg_us, algorithmic_bytes, tb_per_s}] of the launches recorded by _hbm()."""
        torch.cuda.synchronize(self.device)
        out = []
        for (kernel, shape), v in (self.profile_hbm or {}).items():
            ms = sum(r[0].elapsed_time(r[1]) for r in v)
            by = sum(r[2] for r in v)
            out.append({"kernel": kernel, "shape": shape, "launches": len(v), "avg_us": round(ms / len(v) * 1e3, 2),
                        "algorithmic_bytes": int(by / len(v)), "tb_per_s": round(by / (ms * 1e-3) / 1e12, 3) if ms > 0 else None})
        return out

    def collect_profile(self):
        torch.cuda.synchronize(self.device)
        return {k: {"flops": sum(r[2] for r in v), "bytes": sum(r[3] for r in v),
                    "ms": sum(r[0].elapsed_time(r[1]) for r in v), "launches": len(v)}
                for k, v in (self.profile or {}).items()}

    # ------------------------------------------------------------------ resampling
    def warp(self, src: View, flow: View, out: View):
        # (C * 4 * 2 + 8) bytes per pixel: source and result once, the flow once
        self._hbm("warp", f"C{src.C} {src.H}x{src.W}", src.N * src.H * src.W * (8 * src.C + 8),
                  lambda: lib.check(self.L.dcvc_warp(src.ptr, src.cs, flow.ptr, flow.cs, out.ptr, out.cs, src.N, src.H, src.W,
                                                     src.C, self.stream()), "warp"))
        self.calls += 1
        self._rec("warp", src, flow, out)
        return out

    def up2(self, src: View, out: View, scale=1.0, out2: View = None):
        n_in = src.N * src.H * src.W * src.C * 4
        self._hbm("up2", f"C{src.C} {src.H}x{src.W}", n_in * (1 + 4 * (2 if out2 else 1)),
                  lambda: lib.check(self.L.dcvc_up2(src.ptr, src.cs, out.ptr, out.cs, out2.ptr if out2 else None,
                                                    out2.cs if out2 else 0, src.N, src.H, src.W, src.C, scale, self.stream()), "up2"))
        self.calls += 1
        self._rec("up2", src, out, scale, out2)
        return out

    def down2(self, src: View, out: View, scale=1.0, avgpool_order=False):
        self._hbm("down2", f"C{src.C} {src.H}x{src.W}", src.N * src.H * src.W * src.C * 5,
                  lambda: lib.check(self.L.dcvc_down2(src.ptr, src.cs, out.ptr, out.cs, src.N, src.H, src.W, src.C, scale,
                                                      int(avgpool_order), self.stream()), "down2"))
        self.calls += 1
        self._rec_unary("down2", src, out, scale)
        return out

    def maxpool2(self, src: View, out: View):
        lib.check(self.L.dcvc_maxpool2(src.ptr, src.cs, out.ptr, out.cs, src.N, src.H, src.W, src.C, self.stream()),
                  "maxpool2")
        self.calls += 1
        self._rec_unary("maxpool2", src, out)
        return out

    def copy(self, src: View, out: View):
        self._hbm("copy_channels", f"C{src.C} {src.H}x{src.W}", src.N * src.H * src.W * src.C * 8,
                  lambda: lib.check(self.L.dcvc_copy_channels(src.ptr, src.cs, out.ptr, out.cs, src.N * src.H * src.W, src.C,
                                                              self.stream()), "copy_channels"))
        self.calls += 1
        self._rec_unary("copy", src, out)
        return out

    # ------------------------------------------------------------------ SE
    def se_gate(self, name, t: View, w1: torch.Tensor, w2: torch.Tensor, partial=None) -> torch.Tensor:
        """SELayer gate of t.  partial = (buffer, rows, row stride) when the convolution that produced t
        already summed its channels (conv(..., chan_partial=...)): no pass over t at all."""
        N, C_ = t.N, t.C
        mean = self.fbuf(name + ".mean", N * C_)
        gate = self.fbuf(name + ".gate", N * C_)
        if partial is not None:
            buf, parts, stride = partial
            lib.check(self.L.dcvc_channel_mean_finish(buf.data_ptr(), parts, stride, mean.data_ptr(), N, C_, t.HW,
                                                      self.stream()), "channel_mean_finish")
        else:
            scratch = self.fbuf("se_scratch", N * 2048 * 256, scratch=True)
            lib.check(self.L.dcvc_channel_mean(t.ptr, t.cs, mean.data_ptr(), scratch.data_ptr(), N, t.HW, C_,
                                               self.stream()), "channel_mean")
        lib.check(self.L.dcvc_se_gate(mean.data_ptr(), w1.data_ptr(), w2.data_ptr(), gate.data_ptr(), N, C_,
                                      w1.shape[0], self.stream()), "se_gate")
        self.calls += 2
        self._rec("se_gate", t, w1, w2, mean, gate)
        return gate

    # ------------------------------------------------------------------ entropy-model elementwise
    def scale_channels(self, src: View, out: View, q_basic, q_scale, multiply=False, qkey=None):
        lib.check(self.L.dcvc_scale_channels(src.ptr, src.cs, out.ptr, out.cs, q_basic.data_ptr(), q_scale.data_ptr(),
                                             int(multiply), src.N, src.HW, src.C, self.stream()), "scale_channels")
        self.calls += 1
        self._rec("scale_channels", src, out, q_basic, q_scale, multiply, qkey)
        return out

    def round_symbols(self, z: View, z_hat: View, sym: torch.Tensor = None):
        lib.check(self.L.dcvc_round_symbols(z.ptr, z.cs, z_hat.ptr if z_hat else None, z_hat.cs if z_hat else 0,
                                            sym.data_ptr() if sym is not None else None, z.N, z.H, z.W, z.C,
                                            self.stream()), "round_symbols")
        self.calls += 1
        if z_hat is not None:
            self._rec("round", z, z_hat)

    def symbols_to_nhwc(self, sym: torch.Tensor, out: View):
        lib.check(self.L.dcvc_symbols_to_nhwc(sym.data_ptr(), out.ptr, out.cs, out.N, out.H, out.W, out.C,
                                              self.stream()), "symbols_to_nhwc")
        self.calls += 1
        return out

    def dual_prior(self, mode, step, *, y: View = None, fusion: View, spatial: View = None, params: View,
                   y_hat: torch.Tensor, y_q=None, y_res=None, scales_hat=None, sym=None, idx=None, out: View = None,
                   q_basic=None, q_scale=None, distribution="laplace", qkey=None, forced_q=None):
        a = lib.DualPriorArgs()
        Cc = fusion.C // 3
        if y is not None:
            a.y, a.y_cs = y.ptr, y.cs
        a.fusion, a.fusion_cs = fusion.ptr, fusion.cs
        if spatial is not None:
            a.spatial, a.spatial_cs = spatial.ptr, spatial.cs
        a.params, a.params_cs = params.ptr, params.cs
        a.y_hat = y_hat.data_ptr()
        for nm, t in (("y_q", y_q), ("y_res", y_res), ("scales_hat", scales_hat), ("sym", sym), ("idx", idx),
                      ("q_basic", q_basic), ("q_scale", q_scale), ("forced_q", forced_q)):
            if t is not None:
                setattr(a, nm, t.data_ptr())
        if out is not None:
            a.out, a.out_cs = out.ptr, out.cs
        a.N, a.H, a.W, a.C, a.step = fusion.N, fusion.H, fusion.W, Cc, step
        a.idx_edges = self.index_edges(distribution).data_ptr()
        fn = {"enc": self.L.dcvc_dual_prior_enc, "dec_index": self.L.dcvc_dual_prior_dec_index,
              "dec_apply": self.L.dcvc_dual_prior_dec_apply}[mode]
        # operands once each, per element of the (N,H,W,C) planes: step 0 reads y + fusion (3) and writes params (4) +
        # y_hat / symbol / index halves; step 1 reads y + fusion (3) + spatial (2) + y_hat and writes out + the halves
        per_el = {("enc", 0): 4 + 4 + 2.0, ("enc", 1): 7 + 1 + 2.0, ("dec_index", 0): 3 + 3 + 0.5, ("dec_index", 1): 3 + 0.5,
                  ("dec_apply", 0): 3 + 0.5 + 1.5, ("dec_apply", 1): 5 + 0.5 + 1.5}[(mode, step)]
        self._hbm(f"dual_prior_{mode}{step}", f"C{Cc} {fusion.H}x{fusion.W}", fusion.N * fusion.H * fusion.W * Cc * 4 * per_el,
                  lambda: lib.check(fn(C.byref(a), self.stream()), "dual_prior_" + mode))
        self.calls += 1
        if mode == "enc":
            self._rec("dual_prior", step, y, fusion, spatial, params, y_hat, y_res, scales_hat, out, q_basic, q_scale,
                      qkey)

    def index_edges(self, distribution) -> torch.Tensor:
        """Device copy of entropy.scale_index_edges: the 255 fp32 bin edges of build_indexes."""
        t = self._edges.get(distribution)
        if t is None:
            from .entropy import scale_index_edges

            t = scale_index_edges(distribution).to(self.device)
            self._edges[distribution] = t
        return t

    def scale_indexes(self, scales: torch.Tensor, distribution="laplace") -> torch.Tensor:
        """GaussianEncoder.build_indexes (entropy_models.py:264-268) of a flat fp32 device tensor."""
        scales = scales.contiguous()
        idx = torch.empty(scales.numel(), dtype=torch.int32, device=self.device)
        lib.check(self.L.dcvc_scale_indexes(scales.data_ptr(), idx.data_ptr(), scales.numel(),
                                            self.index_edges(distribution).data_ptr(), self.stream()), "scale_indexes")
        self.calls += 1
        return idx.view(scales.shape)

    def _scratch(self, N):
        return self.fbuf("reduce_scratch", N * 1024, scratch=True)

    def scale_bits(self, y_q, scales_hat, N, per_sample, gaussian=False) -> torch.Tensor:
        out = torch.empty(N, dtype=torch.float32, device=self.device)
        lib.check(self.L.dcvc_scale_bits(y_q.data_ptr(), scales_hat.data_ptr(), out.data_ptr(),
                                         self._scratch(N).data_ptr(), int(gaussian), N, per_sample, self.stream()),
                  "scale_bits")
        self.calls += 2
        return out

    def factorized_bits(self, z_hat: View, pblock: torch.Tensor) -> torch.Tensor:
        out = torch.empty(z_hat.N, dtype=torch.float32, device=self.device)
        lib.check(self.L.dcvc_factorized_bits(z_hat.ptr, z_hat.cs, pblock.data_ptr(), out.data_ptr(),
                                              self._scratch(z_hat.N).data_ptr(), z_hat.N, z_hat.HW, z_hat.C,
                                              self.stream()), "factorized_bits")
        self.calls += 2
        return out

    def sq_err(self, a: View, b: View) -> torch.Tensor:
        out = torch.empty(a.N, dtype=torch.float32, device=self.device)
        lib.check(self.L.dcvc_sq_err(a.ptr, a.cs, b.ptr, b.cs, out.data_ptr(), self._scratch(a.N).data_ptr(), a.N,
                                     a.HW, a.C, self.stream()), "sq_err")
        self.calls += 2
        return out
