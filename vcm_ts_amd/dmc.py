"""``DMC`` -- the P-frame codec of DCVC-HEM behind the reference's operator API, running on
the hand-written gfx950 kernels of libdcvc_hip.so.

Drop-in surface (same method names, argument meaning, result keys and state-dict key names
as /root/reference/DCVC_HEM/src/models/video_model.py:131-596 and its base class
src/models/common_model.py:14-217):

    DMC(anchor_num=4)
    .forward_one_frame(x, dpb, mv_y_q_scale=None, y_q_scale=None) / .forward(...)   :470-596
    .compress(x, dpb, mv_y_q_scale, y_q_scale) -> {"dbp", "bit_stream"}              :263-352
    .decompress(dpb, string, height, width, mv_y_q_scale, y_q_scale) -> {"dpb"}      :354-422
    .encode_decode(x, dpb, output_path=None, pic_width=None, pic_height=None, ...)   :424-468
    .update(force=False)                                                             common_model.py:75-80
    DMC.get_q_scales_from_ckpt(path)                                                 :248-253

Differences that are deliberate and documented in DESIGN.md: tensors in the returned DPB
are zero-copy logical-NCHW views over channels-last device buffers that the next call
recycles (two alternating sets; in ``.train()`` mode every call owns fresh buffers instead);
``compress`` also returns the key ``"dpb"`` next to the reference's misspelt ``"dbp"``.

Training: in ``.train()`` mode ``forward_one_frame`` is one torch.autograd node (``_FrameFn``)
whose backward runs the gradient kernels of include/dcvc_hip_grad.h through vcm_ts_amd/grad.py.
"""
from __future__ import annotations

import time

import numpy as np
import torch
from torch import nn

from . import entropy as E
from . import lib
from . import stream as S
from .engine import Engine, View
from .nets import Net
from .params import dmc_spec, seeded_state_dict


class _Holder(nn.Module):
    """Empty container used to rebuild the reference's dotted parameter names."""


def build_param_tree(root: nn.Module, spec, init):
    for name, shape in spec.items():
        parts = name.split(".")
        m = root
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, _Holder())
            m = m._modules[p]
        t = init[name]
        assert tuple(t.shape) == tuple(shape), name
        m.register_parameter(parts[-1], nn.Parameter(t.clone(), requires_grad=False))


class PendingStream:
    """Symbol planes of one picture on their way to the host: the D2H copies were enqueued on
    the launch stream behind the kernels that produce them; finish() waits for them and runs
    the rANS coder.  Lets the caller enqueue the next picture's kernels first (pipeline.py)."""

    def __init__(self, owner, host, event, layout, batch=1):
        self.owner, self.host, self.event, self.layout, self.batch = owner, host, event, layout, batch
        self._streams = None

    def finish_all(self):
        """One payload per batch element: planes are (n, c, y, x) ordered, so element b of every
        plane is a contiguous slice.  (The reference's coder only handles n = 1; a batch of rate
        points -- SURVEY 8f-4 -- simply yields one independent stream per element.)"""
        if self._streams is None:
            self.event.synchronize()
            ec = self.owner.entropy_coder
            flat = self.host.numpy()
            out = []
            for b in range(self.batch):
                ec.reset_encoder()
                for table, s_off, i_off, n, chan in self.layout:
                    per = n // self.batch
                    cdf, ln, off = self.owner._tables[table]
                    lo = s_off + b * per
                    if i_off is None:
                        idx = self.owner._chan_index(1, *chan[1:])
                    else:
                        idx = flat[i_off + b * per : i_off + (b + 1) * per]
                    ec.encode_with_indexes(flat[lo : lo + per], idx, cdf, ln, off)
                out.append(ec.flush_encoder())
            self._streams = out
        return self._streams

    def finish(self) -> bytes:
        return self.finish_all()[0]


class CodecBase(nn.Module):
    """What DMC and IntraNoAR share: parameter tree, engine, q-scale plumbing, tables."""

    _tag = "codec"
    # decompress() clamps the reconstruction to [0, 1] as the reference's does (video_model.py:413, image_model.py:199).
    # tests/test_gpu_codec.py's teacher-forced parity check sets False to follow the reference's ESTIMATE-path DPB
    # recursion (forward_one_frame keeps the unclamped picture, video_model.py:535) through the decoder networks
    _clamp_decoded = True
    # tests: {"mv_z", "z", "mv", "y"} -> integer-valued NCHW tensors (the reference's rounded hyper latents and rounded
    # dual-prior residuals of this picture) that REPLACE this implementation's own roundings in the encoder networks
    # (forced-symbol replay of the gradient fixtures, tests/test_gpu_backward.py).  None in every product path.
    _forced = None

    def __init__(self, spec, seed=0, precision=None):
        super().__init__()
        self.precision = precision  # None -> $DCVC_PRECISION or "fp32" (see include/dcvc_hip.h)
        build_param_tree(self, spec, seeded_state_dict(spec, seed=seed))
        self._pmap = dict(self.named_parameters())
        self._engine = None
        self._net = None
        self.entropy_coder = None
        self._tables = None
        self._flip = 0
        self._chan_cache, self._stage_bufs, self._stage_flip, self._stage_owner = {}, {}, 0, {}
        self._dcoder, self._dc_active, self._dc_stream, self._dc_done = None, False, None, None
        self._graphs = {}
        self._frame_graphs = {}
        self._fork_stream = None

    # -- plumbing ------------------------------------------------------------------------
    def P(self, name):
        return self._pmap[name]

    @property
    def device(self):
        return self._pmap[next(iter(self._pmap))].device

    def engine(self) -> Engine:
        dev = self.device
        if self._engine is None or self._engine.device != dev:
            self._engine = Engine(dev, self.precision)  # raises without a GPU / without libdcvc_hip.so
            self._net = Net(self._engine, self.P, self._tag)
        return self._engine

    def set_precision(self, precision):
        """'fp32' (exact fp32 MFMA, parity mode) or 'fp16x3' (split-fp16 MFMA, fast mode)."""
        self.precision = precision
        self._engine = None
        self._net = None
        return self

    def _qvec(self, q, N, default_param=None):
        """q-scale argument (None | float | 0-d / (N,1,1,1) tensor) -> (N,) fp32 device tensor."""
        if q is None:
            q = self.P(default_param)
        if torch.is_tensor(q):
            q = q.detach().to(device=self.device, dtype=torch.float32).reshape(-1)
            if q.numel() == 1:
                q = q.expand(N)
            assert q.numel() == N, "one q-scale per batch element"
            return q.contiguous()
        return torch.full((N,), float(q), dtype=torch.float32, device=self.device)

    def _frame_in(self, name, x, cs=None) -> View:
        e = self.engine()
        N, C_, H, W = x.shape
        return e.from_nchw(x, e.buf(f"{self._tag}/{name}", N, H, W, C_, cs=cs))

    def _dpb_in(self, name, t) -> View:
        """Reference tensors handed in by the caller: ours are aliased, foreign ones converted."""
        if t is None:
            return None
        v = View.alias(t) if t.device == self.device else None
        return v if v is not None else self._frame_in("in_" + name, t)

    def _out_set(self, *inputs):
        """Pick the DPB buffer set (0/1) that none of the caller's tensors aliases."""
        e = self.engine()
        for k in (self._flip, 1 - self._flip):
            ptrs = {t.data_ptr() for key, t in e.bufs.items() if key[0].startswith(f"{self._tag}/dpb{k}.")}
            if not any(v is not None and v.ptr in ptrs for v in inputs):
                self._flip = 1 - k
                return k
        raise RuntimeError("both DPB buffer sets are aliased by the inputs")

    # -- entropy tables ------------------------------------------------------------------
    _distribution = "laplace"
    _z_names = ("bit_estimator_z",)

    def update(self, force=False, device_tables=False):
        """CompressionModel.update (common_model.py:75-80): build the three CDF tables.  Default: on the host
        with the reference's own fp32 arithmetic (integer-identical tables: interoperable streams).
        device_tables=True: by the GPU kernels (dcvc_build_*_cdfs; an entry may differ by one count from the
        reference's, so only for streams this library both writes and reads)."""
        if self.entropy_coder is not None and not force:
            return
        self.entropy_coder = E.EntropyCoder()
        if device_tables:
            self._tables = E.device_tables(self.engine(), self._distribution, {n: self._zblock(n) for n in self._z_names})
            self._dcoder = None
            return
        sd = {k: v for k, v in self._pmap.items()}
        t = {"scale": E.scale_table_cdfs(self._distribution)}
        for n in self._z_names:
            t[n] = E.factorized_cdfs(E.factorized_params(sd, n))
        self._tables = t

    def _zblock(self, name):
        key = ("zblock", name)
        ver = tuple(self.P(f"{name}.f{i}.h")._version for i in (1, 2, 3, 4))
        c = self.engine().packs.get(key)
        if c is None or c[0] != ver:
            # assembled where the parameters live: while training this runs every step, and a host round trip per
            # parameter (22 synchronising copies per picture) would stall the launch queue
            rows = [self.P(f"{name}.f{i}.{k}") for i in (1, 2, 3) for k in "hba"] + [self.P(f"{name}.f4.h"), self.P(f"{name}.f4.b")]
            blk = torch.stack([r.detach().float().reshape(-1) for r in rows]).contiguous().to(self.device)
            c = (ver, blk)
            self.engine().packs[key] = c
        return c[1]

    # -- host <-> device symbol traffic ---------------------------------------------------
    def _chan_index(self, N, C_, H, W):
        key = (N, C_, H, W)
        c = self._chan_cache.get(key)
        if c is None:
            c = np.ascontiguousarray(np.broadcast_to(np.arange(C_, dtype=np.int32)[None, :, None, None], (N, C_, H, W))).reshape(-1)
            self._chan_cache[key] = c
        return c

    def _stage_symbols(self, planes, batch=1) -> PendingStream:
        """planes: list of (table name, sym int32 device tensor, idx int32 device tensor or None,
        (N, C, H, W) for the per-channel index of factorised planes).  One pinned host buffer per
        alternating slot; copies are asynchronous on the current stream."""
        total = sum(p[1].numel() + (0 if p[2] is None else p[2].numel()) for p in planes)
        slot = self._stage_flip
        self._stage_flip ^= 1
        prev = self._stage_owner.get(slot)
        if prev is not None and prev._streams is None:
            raise RuntimeError("more than two deferred pictures in flight: call pending.finish() on older ones first")
        key = (slot, total)
        host = self._stage_bufs.get(key)
        if host is None:
            host = torch.empty(total, dtype=torch.int32, pin_memory=True)
            self._stage_bufs[key] = host
        layout, off = [], 0
        for table, sym, idx, chan in planes:
            n = sym.numel()
            host[off : off + n].copy_(sym, non_blocking=True)
            s_off, off = off, off + n
            i_off = None
            if idx is not None:
                host[off : off + n].copy_(idx, non_blocking=True)
                i_off, off = off, off + n
            layout.append((table, s_off, i_off, n, chan))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        pending = PendingStream(self, host, ev, layout, batch)
        self._stage_owner[slot] = pending
        return pending

    @staticmethod
    def _stage_layout(planes):
        layout, off = [], 0
        for table, sym, idx, chan in planes:
            n = sym.numel()
            s_off, off = off, off + n
            i_off = None
            if idx is not None:
                i_off, off = off, off + n
            layout.append((table, s_off, i_off, n, chan))
        return layout, off

    # -- opt-in device entropy coder (include/dcvc_hip_rans.h; NOT the reference's wire format) ---
    def device_coder(self) -> "E.DeviceCoder":
        if self._tables is None:
            raise RuntimeError("call update() before compress()/decompress()")
        if self._dcoder is None or self._dcoder.device != self.device or self._dcoder.tables_id != id(self._tables) \
                or self._dcoder.symbols_per_lane != self.device_coder_symbols_per_lane:
            self._dcoder = E.DeviceCoder(self.device, self._tables, symbols_per_lane=self.device_coder_symbols_per_lane)
            self._dcoder.tables_id = id(self._tables)
        return self._dcoder

    device_coder_symbols_per_lane = 512  # see include/dcvc_hip_rans.h: payload size vs coder latency

    def _stage_symbols_device(self, planes, batch=1):
        if batch != 1:
            raise NotImplementedError("the device coder codes one picture per payload (batch 1)")
        dc = self.device_coder()
        # The coder kernels (one workgroup per plane, ~1-2 ms per picture) run on a side stream behind
        # this picture's kernels, so they overlap the next picture's motion estimation; _wait_coder()
        # holds the launch stream back only where the symbol buffers are about to be rewritten.
        main = torch.cuda.current_stream(self.device)
        if self._dc_stream is None:
            self._dc_stream = torch.cuda.Stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        self._dc_stream.wait_event(ready)
        with torch.cuda.stream(self._dc_stream):
            dc.begin()
            for table, sym, idx, chan in planes:
                dc.encode(table, sym, idx, chan=None if idx is not None else (chan[1], chan[2] * chan[3]))
            pending = dc.end()
        self._dc_done = pending.event
        return pending

    fork_features = True  # run the DPB-only feature pyramid on a side stream beside the motion path

    def _fork_pyramid(self, net, dv, tape):
        """Start Net.feature_pyramid on a side stream (inference only: a recorded training forward and
        a graph capture stay single-stream).  Returns (pyramid or None, join event or None)."""
        if not self.fork_features or tape is not None or torch.cuda.is_current_stream_capturing():
            return None, None
        main = torch.cuda.current_stream(self.device)
        if self._fork_stream is None:
            self._fork_stream = torch.cuda.Stream(self.device)
        start = torch.cuda.Event()
        start.record(main)       # everything of the previous picture (readers of these buffers, the DPB) is before this
        self._fork_stream.wait_event(start)
        with torch.cuda.stream(self._fork_stream):
            pyr = net.feature_pyramid(dv["ref_frame"], dv["ref_feature"])
            done = torch.cuda.Event()
            done.record(self._fork_stream)
        return pyr, done

    def _wait_coder(self):
        if self._dc_done is not None:
            torch.cuda.current_stream(self.device).wait_event(self._dc_done)
            self._dc_done = None

    def _encode_factorized(self, name, sym: torch.Tensor, N, C_, H, W):
        cdf, ln, off = self._tables[name]
        s = sym.cpu().numpy()
        idx = np.broadcast_to(np.arange(C_, dtype=np.int32)[None, :, None, None], (N, C_, H, W)).reshape(-1)
        self.entropy_coder.encode_with_indexes(s, idx, cdf, ln, off)

    def _encode_scale(self, sym: torch.Tensor, idx: torch.Tensor):
        cdf, ln, off = self._tables["scale"]
        self.entropy_coder.encode_with_indexes(sym.cpu().numpy(), idx.cpu().numpy(), cdf, ln, off)

    def _decode_factorized(self, name, N, C_, H, W) -> torch.Tensor:
        if self._dc_active:
            return self._dcoder.decode(name, N * C_ * H * W, chan=(C_, H * W))
        cdf, ln, off = self._tables[name]
        idx = np.broadcast_to(np.arange(C_, dtype=np.int32)[None, :, None, None], (N, C_, H, W)).reshape(-1)
        out = self.entropy_coder.decoder.decode_stream(idx, cdf, ln, off)
        return torch.from_numpy(out).to(self.device)

    def _decode_scale(self, idx: torch.Tensor) -> torch.Tensor:
        if self._dc_active:
            return self._dcoder.decode("scale", idx.numel(), idx=idx)
        cdf, ln, off = self._tables["scale"]
        out = self.entropy_coder.decoder.decode_stream(idx.cpu().numpy(), cdf, ln, off)
        return torch.from_numpy(out).to(self.device)

    # -- dual prior (both directions) -----------------------------------------------------
    def _dual_prior_encode(self, tag, y: View, fusion: View, prior_name, out: View, q_basic, q_scale, want_stats,
                           want_symbols, want_res=False, qkey=None):
        """forward_dual_prior (common_model.py:104-177): returns dict with y_q / scales_hat
        (dense NHWC planes for the bit estimate) and the two (sym, idx) int32 pairs."""
        e, net = self.engine(), self._net
        N, H, W, Cc = y.N, y.H, y.W, y.C
        n = N * H * W * Cc
        params = net.buf(f"{tag}.dp_params", like=y, C=4 * Cc)
        y_hat = e.fbuf(f"{self._tag}/{tag}.dp_yhat", n)
        r = {}
        if want_stats:
            r["y_q"] = e.fbuf(f"{self._tag}/{tag}.dp_yq", n)
            r["scales_hat"] = e.fbuf(f"{self._tag}/{tag}.dp_sh", n)
        if want_res:
            r["y_res"] = e.fbuf(f"{self._tag}/{tag}.dp_yres", n)
        sym = [None, None]
        idx = [None, None]
        if want_symbols:
            for k in (0, 1):
                sym[k] = e.ibuf(f"{self._tag}/{tag}.sym{k}", n // 2)
                idx[k] = e.ibuf(f"{self._tag}/{tag}.idx{k}", n // 2)
        forced = None
        if self._forced is not None:  # tests only (teacher forcing, see _forced)
            forced = self._forced[tag].to(device=self.device, dtype=torch.float32).permute(0, 2, 3, 1).contiguous()
            assert tuple(forced.shape) == (N, H, W, Cc), (tag, forced.shape)
        common = dict(y=y, fusion=fusion, params=params, y_hat=y_hat, y_q=r.get("y_q"), scales_hat=r.get("scales_hat"),
                      y_res=r.get("y_res"), distribution=self._distribution, qkey=qkey, forced_q=forced)
        e.dual_prior("enc", 0, sym=sym[0], idx=idx[0], **common)
        spatial = net.three_convs(prior_name, params)
        e.dual_prior("enc", 1, spatial=spatial, sym=sym[1], idx=idx[1], out=out, q_basic=q_basic, q_scale=q_scale,
                     **common)
        r["sym"], r["idx"] = sym, idx
        r["params"], r["spatial"] = params, spatial  # (views, for gradient diagnostics)
        return r

    def _dual_prior_decode(self, tag, fusion: View, prior_name, out: View, q_basic, q_scale):
        """decompress_dual_prior (common_model.py:182-217): two rANS decodes with the spatial
        prior in between; everything else stays on the device."""
        e, net = self.engine(), self._net
        N, H, W = fusion.N, fusion.H, fusion.W
        Cc = fusion.C // 3
        n = N * H * W * Cc
        params = net.buf(f"{tag}.dp_params", N=N, H=H, W=W, C=4 * Cc)
        y_hat = e.fbuf(f"{self._tag}/{tag}.dp_yhat", n)
        idx = e.ibuf(f"{self._tag}/{tag}.idx0", n // 2)
        common = dict(fusion=fusion, params=params, y_hat=y_hat, distribution=self._distribution)
        e.dual_prior("dec_index", 0, idx=idx, **common)
        sym = self._decode_scale(idx)
        e.dual_prior("dec_apply", 0, sym=sym, **common)
        spatial = net.three_convs(prior_name, params)
        e.dual_prior("dec_index", 1, spatial=spatial, idx=idx, **common)
        sym = self._decode_scale(idx)
        e.dual_prior("dec_apply", 1, spatial=spatial, sym=sym, out=out, q_basic=q_basic, q_scale=q_scale, **common)
        return out


class _FrameFn(torch.autograd.Function):
    """One P picture as a single autograd node: forward records a grad.Tape on the HIP engine,
    backward replays it with the kernels of include/dcvc_hip_grad.h.  Inputs after the q-scales
    are all parameters of the model, so optimisers, DDP hooks and requires_grad switches
    (DCVC_HEM.activate_modules_*) work as they do on the reference's nn.Modules; the four DPB
    tensors are inputs and outputs of the node, so a DPB that is not detached carries the
    gradient into the previous picture (the reference's cascade training modes)."""

    DPB_KEYS = ("ref_frame", "ref_feature", "ref_y", "ref_mv_y")

    @staticmethod
    def forward(ctx, model, x, rf, rfeat, ry, rmv, qm, qy, *params):
        from .grad import Tape

        # outputs the caller's loss does not use (the DPB in the `single` modes, rate terms that are switched off) arrive
        # in backward as None instead of as zero tensors that would be transposed into gradient buffers for nothing
        ctx.set_materialize_grads(False)
        tape = Tape(model.engine())
        dpb_in = dict(zip(_FrameFn.DPB_KEYS, (rf, rfeat, ry, rmv)))
        tape.dpb_grad = {k for k, need in zip(_FrameFn.DPB_KEYS, ctx.needs_input_grad[2:6]) if need}
        o, sums = model._train_frame(tape, x.detach(), {k: (None if v is None else v.detach()) for k, v in dpb_in.items()},
                                     qm.detach(), qy.detach())
        ctx.tape, ctx.model, ctx.params = tape, model, params
        ctx.q_shapes = (qm.shape, qy.shape)
        ctx.out_views = (o["recon"], o["feature"], o["y_hat"], o["mv_y_hat"])
        ctx.in_views = tuple(o["dv"][k] for k in _FrameFn.DPB_KEYS)
        d = model._dpb_out(o)
        return (sums["bits_mv_y"], sums["bits_mv_z"], sums["bits_y"], sums["bits_z"], sums["sq"], sums["me_sq"],
                d["ref_frame"], d["ref_feature"], d["ref_y"], d["ref_mv_y"])

    @staticmethod
    def backward(ctx, g_mv_y, g_mv_z, g_y, g_z, g_sq, g_me, *g_dpb):
        tape = ctx.tape
        if tape is None:
            raise RuntimeError("this frame's tape was already consumed (retain_graph is not supported)")
        e = tape.e
        for name, g in (("bits_mv_y", g_mv_y), ("bits_mv_z", g_mv_z), ("bits_y", g_y), ("bits_z", g_z), ("sq", g_sq),
                        ("me_sq", g_me)):
            if g is not None:
                tape.up[name] = g.detach().to(torch.float32).contiguous()
        for v, g in zip(ctx.out_views, g_dpb):  # gradient arriving from the next picture through the DPB
            if g is not None:
                e.from_nchw(g, tape.grad(v))
        tape.backward()
        grads = []
        for p, need in zip(ctx.params, ctx.needs_input_grad[8:]):
            g = tape.pgrads.get(id(p)) if need else None
            grads.append(torch.zeros_like(p) if (need and g is None) else g)

        def qgrad(key, shape, need):
            if not need:
                return None
            g = tape.q[key]["dq_scale"]
            n = 1
            for s_ in shape:
                n *= s_
            return (g.sum() if n == 1 else g.clone()).reshape(shape)  # (a copy: g is a slice of the tape's zeroed pool)

        gqm = qgrad("mv", ctx.q_shapes[0], ctx.needs_input_grad[6])
        gqy = qgrad("y", ctx.q_shapes[1], ctx.needs_input_grad[7])
        g_in = []
        for v, need in zip(ctx.in_views, ctx.needs_input_grad[2:6]):
            gv = tape.grad(v, create=False) if (need and v is not None) else None
            g_in.append(None if gv is None else e.to_nchw(gv))
        ctx.tape = None
        return (None, None, *g_in, gqm, gqy, *grads)


class _FrameGraph:
    """One training picture as TWO captured hipGraphs -- the recorded forward and the tape's reverse pass -- over static
    buffers (round 4).  A batch-4 256x256 step is ~1400 launches of 5-50 us each; issued from Python
    the host needs as long to enqueue them as the GPU to run them (GPU busy 60 % of a step,
    profiles/r03_gpu_timeline_train.txt).  Replayed, the same launches with the same arguments in the same order cost the
    host two calls.  Nothing about the arithmetic changes: the graphs hold exactly the launches _FrameFn would issue
    (test_graphed_training_step_matches_the_eager_one).

    What is static: the picture, the DPB entries, the q-scales and the six upstream gradients are copied into fixed
    buffers before a replay; every activation / gradient buffer the tape allocates lives in the graphs' private pool;
    parameter gradients are slices of one flat buffer (one fill, one copy out); the filters are re-packed from the live
    parameters by the plan launch at the head of the forward graph, so optimiser steps between replays are seen.
    Used only when at most one picture's tape is alive at a time (the `single*` modes: DPB detached) -- DMC._forward_train
    falls back to the eager node otherwise."""

    SUMS = ("bits_mv_y", "bits_mv_z", "bits_y", "bits_z", "sq", "me_sq")

    def __init__(self, model, x, dpb, qm, qy, params):
        from .grad import Tape

        e, dev = model.engine(), model.device
        self.model, self.params, self.busy = model, params, False
        self.x = x.detach().clone(memory_format=torch.contiguous_format)
        self.dpb = {k: (None if v is None else v.detach().clone()) for k, v in dpb.items()}
        self.qm, self.qy = qm.detach().clone(), qy.detach().clone()
        N = x.shape[0]
        self.up = {k: torch.zeros(N, dtype=torch.float32, device=dev) for k in self.SUMS}
        self.offsets, total = {}, 0
        for p in params:
            if p.requires_grad:
                self.offsets[id(p)] = total
                total += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(max(total, 4), dtype=torch.float32, device=dev)

        def forward():
            tape = Tape(e)
            tape.dpb_grad = set()
            tape.parena = (self.flat, self.offsets)
            o, sums = model._train_frame(tape, self.x, self.dpb, self.qm, self.qy)
            return tape, o, sums

        def backward(tape):
            self.flat.zero_()
            tape.up.update(self.up)
            tape.backward()

        # two eager passes first: every lazily created object (packed filters incl. the transposed ones of the data
        # gradient, the packing plan -- rebuilt with a blocking call whenever a pass added filters --, scratch buffers,
        # kernel attributes) must exist before capture, where nothing may synchronise
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(2):
                tape, _, _ = forward()
                backward(tape)
                del tape
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.fwd = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.fwd):
            self.tape, o, sums = forward()
        self.sums = tuple(sums[k] for k in self.SUMS)
        d = model._dpb_out(o)
        self.dpb_out = tuple(d[k] for k in _FrameFn.DPB_KEYS)
        self.bwd = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.bwd, pool=self.fwd.pool()):
            backward(self.tape)
        self.dq = {k: self.tape.q[k]["dq_scale"] for k in ("mv", "y")}

    def load(self, x, dpb, qm, qy):
        self.x.copy_(x)
        for k, v in self.dpb.items():
            if v is not None:
                v.copy_(dpb[k])
        self.qm.copy_(qm)
        self.qy.copy_(qy)


class _GraphedFrameFn(torch.autograd.Function):
    """_FrameFn with the launches replayed from a _FrameGraph: same inputs, same outputs, same gradients."""

    @staticmethod
    def forward(ctx, fg, x, rf, rfeat, ry, rmv, qm, qy, *params):
        ctx.set_materialize_grads(False)
        fg.load(x, dict(zip(_FrameFn.DPB_KEYS, (rf, rfeat, ry, rmv))), qm, qy)
        fg.fwd.replay()
        fg.busy = True
        ctx.fg, ctx.q_shapes = fg, (qm.shape, qy.shape)
        # (copies: the static buffers are overwritten by the next replay, callers keep reconstructions across steps)
        return tuple(t.clone() for t in fg.sums) + tuple(t.clone() for t in fg.dpb_out)

    @staticmethod
    def backward(ctx, *g):
        fg = ctx.fg
        if fg is None or not fg.busy:
            raise RuntimeError("this frame's graph was already consumed (retain_graph is not supported)")
        if any(t is not None for t in g[6:]):
            raise RuntimeError("graph-replayed training pictures carry no gradient through the DPB (detach it, or set "
                               "DMC.graph_training = False for the cascade modes)")
        for name, t in zip(fg.SUMS, g[:6]):
            if t is None:
                fg.up[name].zero_()
            else:
                fg.up[name].copy_(t.detach().reshape(-1))
        fg.bwd.replay()
        fg.busy, ctx.fg = False, None
        out = fg.flat.clone()  # one copy out; the per-parameter gradients are views of it
        grads = [out[fg.offsets[id(p)] : fg.offsets[id(p)] + p.numel()].view(p.shape) if (need and id(p) in fg.offsets) else None
                 for p, need in zip(fg.params, ctx.needs_input_grad[8:])]

        def qgrad(key, shape, need):
            if not need:
                return None
            n = 1
            for s_ in shape:
                n *= s_
            t = fg.dq[key]
            return (t.sum() if n == 1 else t.clone()).reshape(shape)

        return (None, None, None, None, None, None, qgrad("mv", ctx.q_shapes[0], ctx.needs_input_grad[6]),
                qgrad("y", ctx.q_shapes[1], ctx.needs_input_grad[7]), *grads)


class DMC(CodecBase):
    _tag = "dmc"
    _distribution = "laplace"
    _z_names = ("bit_estimator_z", "bit_estimator_z_mv")

    def __init__(self, anchor_num=4, seed=0, precision=None):
        super().__init__(dmc_spec(anchor_num), seed=seed, precision=precision)
        self.DMC_version = "1.19"
        self.anchor_num = int(anchor_num)
        self.channel_mv, self.channel_N, self.channel_M = 64, 64, 96

    @staticmethod
    def get_q_scales_from_ckpt(ckpt_path):
        ckpt = S.get_state_dict(ckpt_path)
        return ckpt["y_q_scale"].reshape(-1), ckpt["mv_y_q_scale"].reshape(-1)

    def get_curr_mv_y_q(self, q_scale):
        """video_model.py:255-257: max(mv_y_q_basic, 0.5) * q_scale (the kernels apply the same product per channel,
        Engine.scale_channels; this accessor is for callers that inspect the step sizes)."""
        return torch.clamp_min(self.P("mv_y_q_basic"), 0.5) * q_scale

    def get_curr_y_q(self, q_scale):
        """video_model.py:259-261."""
        return torch.clamp_min(self.P("y_q_basic"), 0.5) * q_scale

    # ------------------------------------------------------------------ shared analysis
    def _mv_side(self, net: Net, dpb_v, mv_y: View, mv_z_hat: View, N, q_mv, k, mode, decode=False):
        """mv hyper-decoder -> prior fusion -> dual prior -> mv_y_hat (in DPB set k)."""
        ref_mv_y = dpb_v["ref_mv_y"]
        mv_params = net.hyper_dec("mv_hyper_prior_decoder", mv_z_hat)
        if ref_mv_y is None:  # zeros contribute nothing: drop the segment (and its weight slice)
            fusion = net.three_convs("mv_y_prior_fusion", [mv_params], cin_slice=(0, 128))
        else:
            fusion = net.three_convs("mv_y_prior_fusion", [mv_params, ref_mv_y])
        out = net.buf(f"dpb{k}.ref_mv_y", like=mv_params, C=64)
        qb = self.P("mv_y_q_basic").reshape(-1)
        if decode:
            self._dual_prior_decode("mv", fusion, "mv_y_spatial_prior", out, qb, q_mv)
            return out, None
        r = self._dual_prior_encode("mv", mv_y, fusion, "mv_y_spatial_prior", out, qb, q_mv,
                                    want_stats=(mode != "compress"), want_symbols=(mode == "compress"),
                                    want_res=(mode == "train"), qkey="mv")
        return out, r

    def _y_prior(self, net: Net, dpb_v, c3: View, z_hat: View):
        hier = net.hyper_dec("contextual_hyper_prior_decoder", z_hat)
        t = net.conv("temporal_prior_encoder.0", c3, stride=2, out_slope=0.1)
        temporal = net.conv("temporal_prior_encoder.2", t, stride=2)
        ref_y = dpb_v["ref_y"]
        if ref_y is None:
            return net.three_convs("y_prior_fusion", [temporal, hier], cin_slice=(0, 384))
        return net.three_convs("y_prior_fusion", [temporal, hier, ref_y])

    def _views_of_dpb(self, dpb):
        return {k: self._dpb_in(k, dpb.get(k)) for k in ("ref_frame", "ref_feature", "ref_y", "ref_mv_y")}

    def _run(self, x, dpb, mv_y_q_scale, y_q_scale, mode, tape=None):
        """mode 'estimate' (forward_one_frame, unclamped recon as video_model.py:535) or
        'compress' (recon clamped to [0, 1] exactly as the decoder will, :413, so that the
        encoder's own DPB is bit-identical to the decoder's and no decode pass is needed)."""
        e = self.engine()
        net = self._net
        N, _, H, W = x.shape
        assert H % 64 == 0 and W % 64 == 0, "pad to a multiple of 64 first (stream.get_padding_size)"
        q_mv = self._qvec(mv_y_q_scale, N, "mv_y_q_scale")
        q_y = self._qvec(y_q_scale, N, "y_q_scale")
        dv = self._views_of_dpb(dpb)
        k = self._out_set(*dv.values()) if tape is None else 0
        # current frame lives in channels 0-2 of SpyNet's finest 8-channel input buffer
        pyr, pyr_done = self._fork_pyramid(net, dv, tape)
        spy0 = e.buf("dmc/spy.in0", N, H, W, 8)
        x3 = e.from_nchw(x, spy0.slice(0, 3))
        if tape is not None:  # pictures and detached DPB entries carry no gradient
            tape.mark_const(x3)
            for key, v in dv.items():
                if v is not None:
                    tape.keep.append(v.base)
                    if key not in getattr(tape, "dpb_grad", ()):
                        tape.mark_const(v)
        est_mv = net.spynet(x3, dv["ref_frame"])
        mv_y_raw = net.encoder_stack("mv_encoder", est_mv)
        mv_y = e.scale_channels(mv_y_raw, net.buf("mv_y", like=mv_y_raw, C=64), self.P("mv_y_q_basic").reshape(-1), q_mv,
                                qkey="mv")
        mv_z = net.hyper_enc5("mv_hyper_prior_encoder", mv_y)
        mv_z_hat = net.buf("mv_z_hat", like=mv_z, C=64)
        sym_mv_z = e.ibuf("dmc/sym_mv_z", N * 64 * mv_z.HW) if mode == "compress" else None
        self._wait_coder()  # the previous picture's device coder (if any) has read its symbol planes
        e.round_symbols(mv_z, mv_z_hat, sym_mv_z)
        if self._forced is not None:
            e.symbols_to_nhwc(self._forced["mv_z"].to(self.device, torch.int32).contiguous().view(-1), mv_z_hat)
        mv_y_hat, r_mv = self._mv_side(net, dv, mv_y, mv_z_hat, N, q_mv, k, mode)
        mv_hat = net.decoder_stack("mv_decoder", mv_y_hat)
        enc_cat2 = net.buf("enc_cat2", N=N, H=H // 2, W=W // 2, C=128)
        enc_cat3 = net.buf("enc_cat3", N=N, H=H // 4, W=W // 4, C=128)
        if pyr_done is not None:
            torch.cuda.current_stream(self.device).wait_event(pyr_done)
        c1, c2, c3, warp_frame = net.motion_compensation(dv["ref_frame"], dv["ref_feature"], mv_hat, enc_cat2, enc_cat3,
                                                         want_warp_frame=(mode != "compress"), pyramid=pyr)
        y_raw = net.contextual_encoder(x3, c1, enc_cat2, enc_cat3)
        y = e.scale_channels(y_raw, net.buf("y", like=y_raw, C=96), self.P("y_q_basic").reshape(-1), q_y, qkey="y")
        n_ = "contextual_hyper_prior_encoder"
        t = net.conv(f"{n_}.0", y, out_slope=0.01)
        t = net.conv(f"{n_}.2", t, stride=2, out_slope=0.01)
        z = net.conv(f"{n_}.4", t, stride=2)
        z_hat = net.buf("z_hat", like=z, C=64)
        sym_z = e.ibuf("dmc/sym_z", N * 64 * z.HW) if mode == "compress" else None
        e.round_symbols(z, z_hat, sym_z)
        if self._forced is not None:
            e.symbols_to_nhwc(self._forced["z"].to(self.device, torch.int32).contiguous().view(-1), z_hat)
        fusion = self._y_prior(net, dv, c3, z_hat)
        y_hat = net.buf(f"dpb{k}.ref_y", like=y, C=96)
        r_y = self._dual_prior_encode("y", y, fusion, "y_spatial_prior", y_hat, self.P("y_q_basic").reshape(-1), q_y,
                                      want_stats=(mode != "compress"), want_symbols=(mode == "compress"),
                                      want_res=(mode == "train"), qkey="y")
        dec_feature = net.contextual_decoder(y_hat, c2, c3)
        feature = net.buf(f"dpb{k}.ref_feature", N=N, H=H, W=W, C=64)
        recon = net.buf(f"dpb{k}.ref_frame", N=N, H=H, W=W, C=3)
        net.recon_generation(dec_feature, c1, feature, recon, clamp=(mode == "compress"))
        return dict(N=N, H=H, W=W, x3=x3, recon=recon, feature=feature, y_hat=y_hat, mv_y_hat=mv_y_hat,
                    warp_frame=warp_frame, r_mv=r_mv, r_y=r_y, mv_z_hat=mv_z_hat, z_hat=z_hat, sym_mv_z=sym_mv_z,
                    sym_z=sym_z, est_mv=est_mv, mv_hat=mv_hat, c1=c1, c2=c2, c3=c3, y=y, mv_y=mv_y, z=z, mv_z=mv_z,
                    q_mv=q_mv, q_y=q_y, dv=dv, fusion_y=fusion)

    @staticmethod
    def _dpb_out(o):
        return {"ref_frame": o["recon"].nchw(), "ref_feature": o["feature"].nchw(), "ref_y": o["y_hat"].nchw(),
                "ref_mv_y": o["mv_y_hat"].nchw()}

    # ------------------------------------------------------------------ training-mode forward
    _noise_override = None  # tests: {"y", "mv_y", "z", "mv_z"} -> NCHW tensors replacing add_noise's draws

    def _noise(self, key, N, H, W, C_):
        """uniform(-0.5, 0.5) like CompressionModel.add_noise (common_model.py:46-49), dense NHWC."""
        if self._noise_override is not None:
            t = self._noise_override[key].to(device=self.device, dtype=torch.float32)
            assert tuple(t.shape) == (N, C_, H, W), (key, t.shape)
            return t.permute(0, 2, 3, 1).contiguous()
        return torch.empty((N, H, W, C_), dtype=torch.float32, device=self.device).uniform_(-0.5, 0.5)

    def _train_frame(self, tape, x, dpb, mv_y_q_scale, y_q_scale):
        """Recorded forward: everything forward_one_frame computes in training mode
        (video_model.py:470-596 with self.training: straight-through rounding, noisy latents for
        the bit estimate).  Returns per-sample sums (bits_*, squared errors) and the DPB views."""
        e = self.engine()
        e.tape = tape
        try:
            e.repack_all()  # one launch instead of one per layer for every filter the last optimiser step changed
            N = x.shape[0]
            tape.qstate("mv", self.P("mv_y_q_basic"), self._qvec(mv_y_q_scale, N, "mv_y_q_scale"), N, 64)
            tape.qstate("y", self.P("y_q_basic"), self._qvec(y_q_scale, N, "y_q_scale"), N, 96)
            o = self._run(x, dpb, tape.q["mv"]["q_scale"], tape.q["y"]["q_scale"], "train", tape=tape)
            L = e.L
            sums = {}
            sums["sq"] = e.sq_err(o["recon"], o["x3"])
            tape.ops.append(("sq_err", "sq", o["recon"], o["x3"]))
            sums["me_sq"] = e.sq_err(o["warp_frame"], o["x3"])
            tape.ops.append(("sq_err", "me_sq", o["warp_frame"], o["x3"]))
            for name, r, lat, nkey in (("bits_y", o["r_y"], o["y"], "y"), ("bits_mv_y", o["r_mv"], o["mv_y"], "mv_y")):
                per = lat.HW * lat.C
                noise = self._noise(nkey, N, lat.H, lat.W, lat.C)
                y_bit = torch.empty_like(noise).view(-1)
                lib.check(L.dcvc_add_planes(r["y_res"].data_ptr(), lat.C, noise.data_ptr(), lat.C, y_bit.data_ptr(), lat.C,
                                            N * lat.HW, lat.C, e.stream()), "add_planes")
                sums[name] = e.scale_bits(y_bit, r["scales_hat"], N, per)
                tape.ops.append(("scale_bits", name, y_bit, r["scales_hat"], r["y_res"], N, per))
            for name, z, est, nkey in (("bits_z", o["z"], "bit_estimator_z", "z"),
                                       ("bits_mv_z", o["mv_z"], "bit_estimator_z_mv", "mv_z")):
                noise = self._noise(nkey, N, z.H, z.W, z.C)
                z_bit = View(torch.empty_like(noise), z.C)
                lib.check(L.dcvc_add_planes(z.ptr, z.cs, noise.data_ptr(), z.C, z_bit.ptr, z_bit.cs, N * z.HW, z.C,
                                            e.stream()), "add_planes")
                blk = self._zblock(est)
                sums[name] = e.factorized_bits(z_bit, blk)
                plist = [self.P(f"{est}.f{i}.{k}") for i in (1, 2, 3) for k in ("h", "b", "a")]
                plist += [self.P(f"{est}.f4.h"), self.P(f"{est}.f4.b")]
                tape.ops.append(("factorized_bits", name, z_bit, z, blk, plist))
            return o, sums
        finally:
            e.tape = None

    # Replay training pictures from captured hipGraphs (_FrameGraph).  Opt-in: the caller promises the `single` training
    # recursion (one picture's backward before the next picture's forward, DPB detached), which is what trainer.py /
    # trainer_multi.py run in the stages bench.py times; anything else falls back to the eager node by itself.
    graph_training = False

    def _frame_graph(self, x, dpb_t, qm, qy, params):
        """The _FrameGraph for this call, or None when the call cannot be replayed (then the eager _FrameFn runs)."""
        static_noise = self._noise_override is None or all(t.is_cuda and t.dtype == torch.float32 for t in self._noise_override.values())
        if (not static_noise or self._forced is not None or not torch.is_grad_enabled()
                or any(v is not None and v.requires_grad for v in dpb_t) or x.requires_grad
                or torch.cuda.is_current_stream_capturing()):
            return None
        key = (tuple(x.shape), tuple(None if v is None else tuple(v.shape) for v in dpb_t), tuple(qm.shape), tuple(qy.shape),
               tuple((id(p), p.requires_grad, p.data_ptr()) for p in params), self.engine().precision,
               None if self._noise_override is None else tuple(t.data_ptr() for t in self._noise_override.values()))
        fg = self._frame_graphs.get(key)
        if fg is None:
            if len(self._frame_graphs) >= 4:
                self._frame_graphs.pop(next(iter(self._frame_graphs)))
            fg = _FrameGraph(self, x, dict(zip(_FrameFn.DPB_KEYS, dpb_t)), qm, qy, params)
            self._frame_graphs[key] = fg
        return None if fg.busy else fg

    def _forward_train(self, x, dpb, mv_y_q_scale, y_q_scale):
        qm = self.P("mv_y_q_scale") if mv_y_q_scale is None else mv_y_q_scale
        qy = self.P("y_q_scale") if y_q_scale is None else y_q_scale
        qm = qm if torch.is_tensor(qm) else torch.tensor(float(qm), device=self.device)
        qy = qy if torch.is_tensor(qy) else torch.tensor(float(qy), device=self.device)
        # only the parameters this picture uses enter the graph (like the reference, where autograd /
        # DDP's find_unused_parameters see feature_adaptor_I or _P, never both, and a q-scale table
        # only when it is the one passed in)
        skip = ("feature_adaptor_P." if dpb.get("ref_feature") is None else "feature_adaptor_I.")
        params = [p for n, p in self._pmap.items() if not n.startswith(skip) and n not in ("mv_y_q_scale", "y_q_scale")]
        dpb_t = tuple(dpb.get(k) for k in _FrameFn.DPB_KEYS)
        fg = self._frame_graph(x, dpb_t, qm, qy, params) if self.graph_training else None
        if fg is not None:
            outs = _GraphedFrameFn.apply(fg, x, *dpb_t, qm, qy, *params)
        else:
            outs = _FrameFn.apply(self, x, *dpb_t, qm, qy, *params)
        bits_mv_y, bits_mv_z, bits_y, bits_z, sq, me_sq, recon, feature, y_hat, mv_y_hat = outs
        pix = x.shape[2] * x.shape[3]
        bpp_y, bpp_z, bpp_mv_y, bpp_mv_z = bits_y / pix, bits_z / pix, bits_mv_y / pix, bits_mv_z / pix
        bpp = bpp_y + bpp_z + bpp_mv_y + bpp_mv_z
        res = {"bpp_mv_y": bpp_mv_y, "bpp_mv_z": bpp_mv_z, "bpp_y": bpp_y, "bpp_z": bpp_z, "bpp": bpp,
               "me_mse": me_sq / pix, "mse": sq / pix,
               "dpb": {"ref_frame": recon, "ref_feature": feature, "ref_y": y_hat, "ref_mv_y": mv_y_hat}}
        for key, v in (("bit", bpp), ("bit_y", bpp_y), ("bit_z", bpp_z), ("bit_mv_y", bpp_mv_y), ("bit_mv_z", bpp_mv_z)):
            res[key] = torch.sum(v) * pix
        return res

    # ------------------------------------------------------------------ public API
    def forward_one_frame(self, x, dpb, mv_y_q_scale=None, y_q_scale=None):
        if self.training:
            return self._forward_train(x, dpb, mv_y_q_scale, y_q_scale)
        with torch.no_grad():
            return self._forward_eval(x, dpb, mv_y_q_scale, y_q_scale)

    def _forward_eval(self, x, dpb, mv_y_q_scale=None, y_q_scale=None):
        e = self.engine()
        o = self._run(x, dpb, mv_y_q_scale, y_q_scale, "estimate")
        N, pix = o["N"], o["H"] * o["W"]
        mse = e.sq_err(o["x3"], o["recon"]) / pix
        me_mse = e.sq_err(o["x3"], o["warp_frame"]) / pix
        per_y = o["y"].HW * 96
        per_mv = o["mv_y"].HW * 64
        bpp_y = e.scale_bits(o["r_y"]["y_q"], o["r_y"]["scales_hat"], N, per_y) / pix
        bpp_mv_y = e.scale_bits(o["r_mv"]["y_q"], o["r_mv"]["scales_hat"], N, per_mv) / pix
        bpp_z = e.factorized_bits(o["z_hat"], self._zblock("bit_estimator_z")) / pix
        bpp_mv_z = e.factorized_bits(o["mv_z_hat"], self._zblock("bit_estimator_z_mv")) / pix
        bpp = bpp_y + bpp_z + bpp_mv_y + bpp_mv_z
        res = {"bpp_mv_y": bpp_mv_y, "bpp_mv_z": bpp_mv_z, "bpp_y": bpp_y, "bpp_z": bpp_z, "bpp": bpp,
               "me_mse": me_mse, "mse": mse, "dpb": self._dpb_out(o)}
        for key, v in (("bit", bpp), ("bit_y", bpp_y), ("bit_z", bpp_z), ("bit_mv_y", bpp_mv_y), ("bit_mv_z", bpp_mv_z)):
            res[key] = torch.sum(v) * pix
        res["_views"] = o
        return res

    def forward(self, x, dpb, mv_y_q_scale=None, y_q_scale=None):
        return self.forward_one_frame(x, dpb, mv_y_q_scale=mv_y_q_scale, y_q_scale=y_q_scale)

    def _compress_graph(self, x, dpb, mv_y_q_scale, y_q_scale):
        """compress() as a hipGraph replay.  A P picture is ~235 launches; at small picture sizes the
        host cannot enqueue them as fast as the GPU runs them (256x256: 2.8 ms of enqueue for ~1.5 ms
        of kernels), so the launch sequence is captured once per (picture size, q-scales, DPB buffer
        set) and replayed.  Everything a replay touches is static: the picture is copied into a fixed
        input buffer first, the DPB alternates between the engine's two output sets (one graph each),
        the symbol planes land in a pinned host buffer owned by the graph."""
        if not all(isinstance(q, (int, float)) for q in (mv_y_q_scale, y_q_scale)):
            raise TypeError("graph replay needs plain float q-scales (they are baked into the captured launches)")
        if x.shape[0] != 1:
            raise NotImplementedError("graph replay codes one picture per call")
        ptrs = tuple(None if dpb.get(k) is None else (dpb[k].data_ptr(), tuple(dpb[k].shape), tuple(dpb[k].stride()))
                     for k in ("ref_frame", "ref_feature", "ref_y", "ref_mv_y"))
        key = (tuple(x.shape), ptrs, self._flip, float(mv_y_q_scale), float(y_q_scale))
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 8:
                self._graphs.pop(next(iter(self._graphs)))
            xs = torch.empty_like(x, memory_format=torch.contiguous_format)
            xs.copy_(x)
            flip0 = self._flip
            # eager warm-up with the same state: packs weights, allocates every buffer (nothing may be
            # allocated while capturing); its results are overwritten by the identical captured run
            o = self._run(xs, dpb, mv_y_q_scale, y_q_scale, "compress")
            zm, zz = o["mv_z_hat"], o["z_hat"]
            planes = lambda o: [("bit_estimator_z_mv", o["sym_mv_z"], None, (1, 64, zm.H, zm.W)),
                                ("scale", o["r_mv"]["sym"][0], o["r_mv"]["idx"][0], None),
                                ("scale", o["r_mv"]["sym"][1], o["r_mv"]["idx"][1], None),
                                ("bit_estimator_z", o["sym_z"], None, (1, 64, zz.H, zz.W)),
                                ("scale", o["r_y"]["sym"][0], o["r_y"]["idx"][0], None),
                                ("scale", o["r_y"]["sym"][1], o["r_y"]["idx"][1], None)]
            layout, total = self._stage_layout(planes(o))
            host = torch.empty(total, dtype=torch.int32, pin_memory=True)
            torch.cuda.synchronize(self.device)
            self._flip = flip0
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                o = self._run(xs, dpb, mv_y_q_scale, y_q_scale, "compress")
                for (table, s_off, i_off, n, chan), (_, sym, idx, _) in zip(layout, planes(o)):
                    host[s_off : s_off + n].copy_(sym, non_blocking=True)
                    if idx is not None:
                        host[i_off : i_off + n].copy_(idx, non_blocking=True)
            g = dict(graph=graph, xs=xs, host=host, layout=layout, out=self._dpb_out(o), views=o, flip_after=self._flip,
                     pending=None)
            self._graphs[key] = g
        prev = g["pending"]
        if prev is not None and prev._streams is None:
            raise RuntimeError("this graph's previous picture has not been retired: call pending.finish() first")
        g["xs"].copy_(x)
        g["graph"].replay()
        self._flip = g["flip_after"]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        g["pending"] = PendingStream(self, g["host"], ev, g["layout"], 1)
        return g

    @torch.no_grad()
    def compress(self, x, dpb, mv_y_q_scale, y_q_scale, defer=False, coder="host", graph=False, check_range=True):
        """defer=True returns {"dpb", "pending"}: call pending.finish() later for the bytes.
        check_range: a call that returns bytes (defer=False) raises lib.KernelError if a split-fp16 kernel met an
        activation beyond +-8188 (one status read after the picture is done); pipelined callers (defer=True, or
        check_range=False) read Engine.check_status() / status_snapshot() themselves, as GopEncoder does per GOP.
        coder="device": opt-in lane-interleaved GPU coder (include/dcvc_hip_rans.h, its own format).
        graph=True: replay the picture's launches as a captured hipGraph (host coder, batch 1, float
        q-scales): pays when the picture is small enough for the host enqueue to be the bottleneck."""
        if self.entropy_coder is None:
            raise RuntimeError("call update() before compress()/decompress()")
        if graph:
            if coder != "host":
                raise NotImplementedError("graph replay is wired to the host coder")
            g = self._compress_graph(x, dpb, mv_y_q_scale, y_q_scale)
            d = g["out"]
            if defer:
                return {"dbp": d, "dpb": d, "pending": g["pending"], "_views": g["views"]}
            streams = g["pending"].finish_all()
            if check_range:
                self.engine().check_status()
            return {"dbp": d, "dpb": d, "bit_stream": streams[0], "bit_streams": streams, "_views": g["views"]}
        o = self._run(x, dpb, mv_y_q_scale, y_q_scale, "compress")
        N = o["N"]  # N > 1: a batch of rate points, one independent stream per element ("bit_streams")
        zm, zz = o["mv_z_hat"], o["z_hat"]
        # bitstream order: mv_z, mv_y step 0, mv_y step 1, z, y step 0, y step 1 (video_model.py:333-340)
        assert coder in ("host", "device")
        pending = (self._stage_symbols if coder == "host" else self._stage_symbols_device)([
            ("bit_estimator_z_mv", o["sym_mv_z"], None, (N, 64, zm.H, zm.W)),
            ("scale", o["r_mv"]["sym"][0], o["r_mv"]["idx"][0], None),
            ("scale", o["r_mv"]["sym"][1], o["r_mv"]["idx"][1], None),
            ("bit_estimator_z", o["sym_z"], None, (N, 64, zz.H, zz.W)),
            ("scale", o["r_y"]["sym"][0], o["r_y"]["idx"][0], None),
            ("scale", o["r_y"]["sym"][1], o["r_y"]["idx"][1], None),
        ], batch=N)
        d = self._dpb_out(o)
        if defer:
            return {"dbp": d, "dpb": d, "pending": pending, "_views": o}
        streams = pending.finish_all()
        if check_range:
            self.engine().check_status()
        return {"dbp": d, "dpb": d, "bit_stream": streams[0], "bit_streams": streams, "_views": o}

    @torch.no_grad()
    def decompress(self, dpb, string, height, width, mv_y_q_scale, y_q_scale, coder=None, defer_check=False,
                   check_range=True):
        """coder: "host" (reference format), "device" (payloads of compress(coder="device")) or None =
        tell them apart by the device format's magic.  defer_check (device format only): do not
        synchronise to read the kernels' status word; the caller calls device_coder().check() later."""
        self._defer_check = defer_check
        if self.entropy_coder is None:
            raise RuntimeError("call update() before compress()/decompress()")
        if coder is None:
            coder = "device" if string[:4] == E.DRANS_MAGIC else "host"
        self._dc_active = coder == "device"
        try:
            r = self._decompress(dpb, string, height, width, mv_y_q_scale, y_q_scale)
        finally:
            self._dc_active = False
        if check_range:  # (see compress)
            self.engine().check_status()
        return r

    def _decompress(self, dpb, string, height, width, mv_y_q_scale, y_q_scale):
        e = self.engine()
        net = self._net
        N = 1
        q_mv = self._qvec(mv_y_q_scale, N, "mv_y_q_scale")
        q_y = self._qvec(y_q_scale, N, "y_q_scale")
        dv = self._views_of_dpb(dpb)
        k = self._out_set(*dv.values())
        if self._dc_active:
            self.device_coder().set_stream(string)
        else:
            self.entropy_coder.set_stream(string)
        pyr, pyr_done = self._fork_pyramid(net, dv, None)
        zh, zw = S.get_downsampled_shape(height, width, 64)
        H, W = zh * 64, zw * 64
        sym = self._decode_factorized("bit_estimator_z_mv", N, 64, zh, zw)
        mv_z_hat = e.symbols_to_nhwc(sym, net.buf("mv_z_hat", N=N, H=zh, W=zw, C=64))
        mv_y_hat, _ = self._mv_side(net, dv, None, mv_z_hat, N, q_mv, k, "decode", decode=True)
        mv_hat = net.decoder_stack("mv_decoder", mv_y_hat)
        enc_cat2 = net.buf("enc_cat2", N=N, H=H // 2, W=W // 2, C=128)
        enc_cat3 = net.buf("enc_cat3", N=N, H=H // 4, W=W // 4, C=128)
        if pyr_done is not None:
            torch.cuda.current_stream(self.device).wait_event(pyr_done)
        c1, c2, c3, _ = net.motion_compensation(dv["ref_frame"], dv["ref_feature"], mv_hat, enc_cat2, enc_cat3, False, pyramid=pyr)
        sym = self._decode_factorized("bit_estimator_z", N, 64, zh, zw)
        z_hat = e.symbols_to_nhwc(sym, net.buf("z_hat", N=N, H=zh, W=zw, C=64))
        fusion = self._y_prior(net, dv, c3, z_hat)
        y_hat = net.buf(f"dpb{k}.ref_y", N=N, H=H // 16, W=W // 16, C=96)
        self._dual_prior_decode("y", fusion, "y_spatial_prior", y_hat, self.P("y_q_basic").reshape(-1), q_y)
        dec_feature = net.contextual_decoder(y_hat, c2, c3)
        feature = net.buf(f"dpb{k}.ref_feature", N=N, H=H, W=W, C=64)
        recon = net.buf(f"dpb{k}.ref_frame", N=N, H=H, W=W, C=3)
        net.recon_generation(dec_feature, c1, feature, recon, clamp=self._clamp_decoded)  # recon.clamp(0, 1), :413
        if self._dc_active:
            self._dcoder.release()
            if not self._defer_check:
                self._dcoder.check()  # the one synchronisation of a device-coded picture
        o = dict(recon=recon, feature=feature, y_hat=y_hat, mv_y_hat=mv_y_hat)
        return {"dpb": self._dpb_out(o)}

    def encode_decode(self, x, dpb, output_path=None, pic_width=None, pic_height=None, mv_y_q_scale=None,
                      y_q_scale=None):
        if output_path is not None:
            mv_y_q_scale, mv_y_q_index = S.get_rounded_q(mv_y_q_scale)
            y_q_scale, y_q_index = S.get_rounded_q(y_q_scale)
            t0 = time.time()
            encoded = self.compress(x, dpb, mv_y_q_scale, y_q_scale)
            S.encode_p(encoded["bit_stream"], mv_y_q_index, y_q_index, output_path)
            bits = S.filesize(output_path) * 8
            t1 = time.time()
            mv_y_q_index, y_q_index, string = S.decode_p(output_path)
            decoded = self.decompress(dpb, string, pic_height, pic_width, mv_y_q_index / 100, y_q_index / 100, coder="host")
            torch.cuda.synchronize(self.device)
            t2 = time.time()
            return {"dpb": decoded["dpb"], "bit": bits, "encoding_time": t1 - t0, "decoding_time": t2 - t1}
        enc = self.forward_one_frame(x, dpb, mv_y_q_scale=mv_y_q_scale, y_q_scale=y_q_scale)
        return {"dpb": enc["dpb"], "bit_y": enc["bit_y"].item(), "bit_z": enc["bit_z"].item(),
                "bit_mv_y": enc["bit_mv_y"].item(), "bit_mv_z": enc["bit_mv_z"].item(), "bit": enc["bit"].item(),
                "decoding_time": 0}
