"""Reverse pass over a recorded forward of the HIP engine (training).

A ``Tape`` is attached to the engine for one training-mode ``forward_one_frame``: every engine
call appends what it did (engine.Engine._rec), buffers come from the tape's own arena instead of
the recycled inference workspace, and ``Tape.backward`` walks the list in reverse launching the
gradient kernels of include/dcvc_hip_grad.h.  This replaces torch.autograd *inside* the frame
(the reference differentiates DMC.forward_one_frame, DCVC_HEM/src/models/video_model.py:470-596,
through ATen); outside the frame -- loss assembly, optimiser, DDP hooks -- the caller's torch
code is unchanged because the frame is exposed as one torch.autograd.Function (dmc.py).

Conventions: a gradient buffer mirrors the layout of the forward buffer it belongs to (same
strided-NHWC geometry, keyed by the forward base tensor), is zero-initialised on first use and
every backward launch *accumulates* into its input gradients, so fan-out (residuals, concat
slices consumed by several layers) needs no bookkeeping.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib
from .engine import Engine, View, _r4

WGRAD_SCRATCH_FLOATS = 24 * 1024 * 1024


class Tape:
    def __init__(self, engine: Engine):
        self.e = engine
        self.L = engine.L
        self.ops = []
        self.arena = {}
        self.gbufs = {}       # forward base data_ptr -> flat zero-initialised gradient tensor
        self.keep = []        # forward tensors that must outlive the arena dict (aliases of caller tensors)
        self.const = set()    # (ptr, C) of views that need no gradient (input pictures)
        self.vec = {}         # data_ptr of small forward tensors (SE gates) -> gradient tensor
        self.dense = {}       # data_ptr of dense forward planes (y_res, scales_hat) -> gradient tensor
        self.pgrads = {}      # id(parameter) -> gradient tensor
        self.q = {}           # qkey -> dict(q_basic param, q_scale tensor, dq_mul, s_div)
        self.up = {}          # name of a per-sample output sum -> upstream gradient (N,) tensor
        self.pool = []        # zero-filled chunks the gradient buffers are carved from (one memset each)
        self.pool_left = 0
        self.parena = None    # (flat zero-filled tensor, {id(parameter): offset}) when parameter gradients share one buffer
        self.small, self.small_left = None, 0  # zeroed chunk for the small parameter gradients (pgrad)
        self.fused_dpre = {}  # forward base data_ptr of a convolution output -> its pre-activation gradient, already
                              # computed by the data-gradient launch of its only consumer (round 4, _plan_fusion)
        self.fuse = {}        # forward base data_ptr -> the producing conv op, for outputs eligible for that fusion

    # ------------------------------------------------------------------ bookkeeping
    def mark_const(self, v: View):
        if v is not None:
            self.const.add((v.ptr, v.C))

    def is_const(self, v: View):
        return (v.ptr, v.C) in self.const

    def grad(self, v: View, create=True):
        key = v.base.data_ptr()
        g = self.gbufs.get(key)
        if g is None:
            if not create:
                return None
            g = self.zeros(v.N * v.H * v.W * v.cs)
            self.gbufs[key] = g
        off = (v.ptr - key) // 4
        return View(g, v.C, 0, geom=(v.N, v.H, v.W, v.cs, g.data_ptr() + 4 * off))

    POOL_CHUNK = 64 * 1024 * 1024  # floats

    def zeros(self, n):
        """n zero floats, 16-byte aligned, from the current chunk (hundreds of gradient buffers per
        picture would otherwise cost one fill launch each)."""
        n4 = (n + 3) // 4 * 4
        if n4 > self.pool_left:
            size = max(self.POOL_CHUNK, n4)
            self.pool.append(torch.zeros(size, dtype=torch.float32, device=self.e.device))
            self.pool_left = size
        chunk = self.pool[-1]
        off = chunk.numel() - self.pool_left
        self.pool_left -= n4
        return chunk[off : off + n]

    def pgrad(self, p: torch.Tensor, zero=True):
        """zero=False: the caller's kernel writes (=) every element, so no fill launch is spent."""
        g = self.pgrads.get(id(p))
        if g is None:
            if self.parena is not None and id(p) in self.parena[1]:
                # graph mode (dmc._FrameGraph): every parameter gradient is a slice of ONE flat buffer that the backward
                # graph zeroes with a single fill and hands out with a single copy
                off = self.parena[1][id(p)]
                g = self.parena[0][off : off + p.numel()].view(p.shape)
            elif zero and p.numel() <= 4096:
                # small accumulated gradients (factorised-prior and SE parameters, q_basic: ~100 per step) share one
                # dedicated zeroed chunk instead of costing a fill launch each; the chunk lives as long as those
                # gradients do (it is NOT the pool of activation gradients: that one must die with the tape)
                if self.small_left < p.numel():
                    self.small = torch.zeros(1 << 18, dtype=torch.float32, device=self.e.device)
                    self.small_left = self.small.numel()
                off = self.small.numel() - self.small_left
                n4 = (p.numel() + 3) // 4 * 4
                self.small_left -= n4
                g = self.small[off : off + p.numel()].view(p.shape)
            else:
                g = (torch.zeros_like if zero else torch.empty_like)(p, memory_format=torch.contiguous_format)
            self.pgrads[id(p)] = g
        return g

    def wants(self, p):
        return p is not None and p.requires_grad

    def stream(self):
        return self.e.stream()

    def scratch(self, n):
        return self.e.fbuf("grad_scratch", n, scratch=True)

    def side_stream(self):
        if not getattr(self.e, "wgrad_side_stream", True):  # (developer A/B: everything on the launch stream)
            return torch.cuda.current_stream(self.e.device)
        if getattr(self.e, "_wgrad_stream", None) is None:
            self.e._wgrad_stream = torch.cuda.Stream(self.e.device)
        self.used_side = True
        return self.e._wgrad_stream

    # ------------------------------------------------------------------ kernels
    def channel_dot(self, a: View, b, out: torch.Tensor, over_batch, accumulate):
        sc = self.scratch(a.N * 256 * _r4(max(a.C, 4)))
        lib.check(self.L.dcvc_channel_dot(a.ptr, a.cs, b.ptr if b is not None else None, b.cs if b is not None else 0,
                                          out.data_ptr(), sc.data_ptr(), a.N, a.HW, a.C, int(over_batch),
                                          int(accumulate), self.stream()), "channel_dot")

    def accumulate(self, src: View, dst: View, mask_x: View = None, slope=0.0):
        lib.check(self.L.dcvc_mask_accumulate(src.ptr, src.cs, mask_x.ptr if mask_x is not None else None,
                                              mask_x.cs if mask_x is not None else 0, float(slope), dst.ptr, dst.cs,
                                              src.N * src.HW, src.C, self.stream()), "mask_accumulate")

    # ------------------------------------------------------------------ reverse pass
    def backward(self):
        e = self.e
        saved, e.tape = e.tape, None  # the gradient launches themselves are not recorded
        guard, e.guard_outputs = e.guard_outputs, False  # the data-gradient launches reuse the forward kernel on gradients
        try:
            self._plan_fusion()
            for op in reversed(self.ops):
                getattr(self, "_b_" + op[0])(*op[1:])
            for qk, q in self.q.items():
                self._q_finish(q)
            if getattr(self, "used_side", False):  # parameter gradients are complete for the caller's stream
                torch.cuda.current_stream(e.device).wait_stream(e._wgrad_stream)
        finally:
            e.tape = saved
            e.guard_outputs = guard

    # -- convolution ----------------------------------------------------------------------
    epilogue_fusion = 3  # bit 0: single-consumer LeakyReLU layers, bit 1: activation-on-load layers (developer A/B: 0 = one
                         # epilogue-backward / mask-accumulate launch per layer, as rounds 1-3)

    def _plan_fusion(self):
        """Which convolution outputs get their epilogue backward from their consumer (round 4).  A layer whose epilogue is
        a LeakyReLU / ReLU only and whose output feeds exactly ONE later convolution (whole tensor, no activation on load)
        needs dpre = dout * act'(out), and dout is produced by nothing but that consumer's data-gradient launch: the
        launch applies the mask in its epilogue (dcvc_conv_args.out_act 3, mask source = the forward activation) and
        writes dpre directly -- no zero-filled gradient buffer, no accumulate read, no epilogue-backward launch.  The
        values are the same bit for bit (0 + v, then v * m).  Eligibility is decided by counting every appearance of a
        forward buffer among the recorded operations' arguments: producer + one consumer = 2."""
        self.fuse, self.fused_dpre = {}, {}
        if not (self.epilogue_fusion & 1):
            return
        uses, producer = {}, {}

        def walk(a):
            if isinstance(a, View):
                k = a.base.data_ptr()
                uses[k] = uses.get(k, 0) + 1
            elif isinstance(a, (tuple, list)):
                for x in a:
                    walk(x)
            elif isinstance(a, dict):
                for x in a.values():
                    walk(x)

        for op in self.ops:
            walk(op[1:])
            if op[0] == "conv":
                pk, srcs, out, stride, in_slope, out_slope, res, gate, res2 = op[1:]
                if (out_slope is not None and out_slope != "clamp01" and res is None and res2 is None and gate is None
                        and not pk.ps and stride == 1):
                    producer[out.base.data_ptr()] = op
        for k, op in producer.items():
            if uses.get(k) == 2 and k not in self.gbufs:
                self.fuse[k] = op

    def _fusable(self, s: View, in_slope):
        """The producing conv op if this source view's gradient can be written as the producer's dpre, else None."""
        if in_slope is not None or self.is_const(s):
            return None
        op = self.fuse.get(s.base.data_ptr())
        if op is None:
            return None
        out = op[3]
        if (out.ptr, out.C, out.cs, out.N, out.H, out.W) != (s.ptr, s.C, s.cs, s.N, s.H, s.W):
            return None
        return op

    def _b_conv(self, pk, srcs, out, stride, in_slope, out_slope, res, gate, res2):
        e, L = self.e, self.L
        fused = self.fused_dpre.pop(out.base.data_ptr(), None)
        dout = None if fused is not None else self.grad(out, create=False)
        if dout is None and fused is None:
            return
        s0 = srcs[0]
        ks = pk.ks
        pad = ks // 2
        N, Ho, Wo = s0.N, (s0.H + 2 * pad - ks) // stride + 1, (s0.W + 2 * pad - ks) // stride + 1
        Cout = pk.Cout
        act = out_slope is not None
        assert out_slope != "clamp01", "clamped outputs are not differentiable (compress mode only)"
        need_pro = act or res is not None or res2 is not None or pk.ps or stride == 2
        if fused is not None:  # the consumer's data-gradient launch already wrote dpre (see _plan_fusion)
            zs, Hd, Wd = 1, Ho, Wo
            dpre, need_pro = fused, False
        elif need_pro:
            zs = stride
            Hd, Wd = (s0.H, s0.W) if zs == 2 else (Ho, Wo)
            t = (torch.zeros if zs == 2 else torch.empty)((N, Hd, Wd, _r4(Cout)), dtype=torch.float32, device=e.device)
            dpre = View(t, Cout)
            a = lib.ConvBwdArgs()
            a.dout, a.dout_cs = dout.ptr, dout.cs
            a.out, a.out_cs = out.ptr, out.cs
            if res is not None:
                a.res, a.res_cs = res.ptr, res.cs
                if not self.is_const(res):
                    dres = self.grad(res)
                    a.dres, a.dres_cs = dres.ptr, dres.cs
            if gate is not None:
                a.gate = gate.data_ptr()
            if res2 is not None:
                a.res2, a.res2_cs = res2.ptr, res2.cs
                if not self.is_const(res2):
                    d2 = self.grad(res2)
                    a.dres2, a.dres2_cs = d2.ptr, d2.cs
            a.dpre, a.dpre_cs, a.zs, a.Hd, a.Wd = dpre.ptr, dpre.cs, zs, Hd, Wd
            a.N, a.Ho, a.Wo, a.Cout = N, Ho, Wo, Cout
            a.pixel_shuffle, a.act, a.slope = int(pk.ps), int(act), float(out_slope or 0.0)
            lib.check(L.dcvc_conv_bwd_prologue(C.byref(a), self.stream()), "conv_bwd_prologue")
        else:
            zs, Hd, Wd = 1, Ho, Wo
            dpre = dout
        if gate is not None:  # d gate(n, c) = sum_pixels dout * res
            dg = self.vec.get(gate.data_ptr())
            if dg is None:
                dg = self.zeros(gate.numel()).view(gate.shape)  # (from the zeroed pool: no fill launch of its own)
                self.vec[gate.data_ptr()] = dg
            self.channel_dot(dout, res, dg, over_batch=False, accumulate=True)
        off = 0 if pk.cin_slice is None else pk.cin_slice[0]
        cin_total = pk.weight.shape[1]
        fused_bias = self.wants(pk.bias) and self.wants(pk.weight) and id(pk.bias) not in self.pgrads
        if self.wants(pk.bias) and not fused_bias:
            self.channel_dot(dpre, None, self.pgrad(pk.bias), over_batch=True, accumulate=True)
        if self.wants(pk.weight):
            # a layer whose segments cover every input channel writes its whole gradient: no zero fill
            whole = sum(s.C for s in srcs) == cin_total and id(pk.weight) not in self.pgrads
            dw = self.pgrad(pk.weight, zero=not whole)
            db = self.pgrad(pk.bias, zero=False) if fused_bias else None
            sc = e.fbuf("wgrad_scratch", WGRAD_SCRATCH_FLOATS, scratch=True)
            # Weight gradients are needed only at the end of the pass and depend on nothing but dpre
            # and the stored activations: they run on a second stream beside the data-gradient chain
            # (the only chain the next layer waits for), so the two kinds of launches fill each
            # other's idle CUs.
            main = torch.cuda.current_stream(e.device)
            side = self.side_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            if need_pro or fused is not None:
                if torch.cuda.is_current_stream_capturing():
                    self.keep.append(dpre.base)  # (a captured pass keeps its temporaries: the graph owns them anyway)
                else:
                    dpre.base.record_stream(side)  # a temporary: keep the allocator from recycling it early
            wstream = C.c_void_p(side.cuda_stream)
            o = off
            for si, s in enumerate(srcs):
                w = lib.WgradArgs()
                w.x, w.x_cs, w.C = s.ptr, s.cs, s.C
                w.in_act, w.in_slope = (0, 0.0) if in_slope is None else (1, float(in_slope))
                w.dpre, w.dpre_cs, w.zs, w.Hd, w.Wd = dpre.ptr, dpre.cs, zs, Hd, Wd
                w.N, w.Hin, w.Win, w.Ho, w.Wo, w.Cout, w.ks, w.stride = N, s.H, s.W, Ho, Wo, Cout, ks, stride
                w.dw, w.Cin_total, w.cin_offset = dw.data_ptr(), cin_total, o
                w.scratch, w.scratch_floats = sc.data_ptr(), sc.numel()
                w.overwrite = int(whole)
                w.precision = lib.PRECISIONS[e.precision] if e.wgrad_split else lib.PRECISIONS["fp32"]
                if db is not None and si == 0:  # the bias gradient rides on the first segment's pass over dY
                    w.db = db.data_ptr()
                lib.check(L.dcvc_conv_wgrad(C.byref(w), wstream), "conv_wgrad")
                o += s.C
        # data gradient: the forward kernel on the flipped / transposed filter, one launch per segment
        o = off
        dsrc_in = View(dpre.base, Cout, 0, geom=(N, Hd, Wd, dpre.cs, dpre.ptr))
        for s in srcs:
            if not self.is_const(s):
                pkT = e.pack_dev((pk.key, "T", o), pk.weight, None, (s.C,), False, cin_slice=(o, o + s.C), transposed=True)
                prod = self._fusable(s, in_slope)
                if prod is not None and s.base.data_ptr() not in self.gbufs:
                    # s is the activated output of `prod` and this launch is its only gradient source: write the
                    # producer's dpre = conv * act'(s) directly (mask source: the forward activation itself)
                    dp = View(torch.empty((s.N, s.H, s.W, _r4(s.C)), dtype=torch.float32, device=e.device), s.C)
                    e.conv(pkT, [dsrc_in], dp, out_slope=("mask", float(prod[6])), res2=s)
                    self.fused_dpre[s.base.data_ptr()] = dp
                    o += s.C
                    continue
                ds = self.grad(s)
                if in_slope is None:
                    e.conv(pkT, [dsrc_in], ds, res=ds)  # in place: ds += conv
                elif self.epilogue_fusion & 2:
                    # activation on load in the forward: ds += conv * act'(s), in the launch's own epilogue (out_act 3 with
                    # a residual: one fma, as dcvc_mask_accumulate computes it)
                    e.conv(pkT, [dsrc_in], ds, res=ds, out_slope=("mask", float(in_slope)), res2=s)
                else:
                    tmp = View(torch.empty((s.N, s.H, s.W, _r4(s.C)), dtype=torch.float32, device=e.device), s.C)
                    e.conv(pkT, [dsrc_in], tmp)
                    self.accumulate(tmp, ds, mask_x=s, slope=in_slope)
            o += s.C

    # -- resampling -----------------------------------------------------------------------
    def _b_warp(self, src, flow, out):
        dout = self.grad(out, create=False)
        if dout is None:
            return
        dsrc = None if self.is_const(src) else self.grad(src)
        dflow = None if self.is_const(flow) else self.grad(flow)
        if dsrc is None and dflow is None:
            return
        fix = None
        if dsrc is not None:  # zeroed once; the kernels hand it back clean
            n = src.N * src.H * src.W * src.C + 1  # + the call's control word (dcvc_hip_grad.h)
            fix = getattr(self.e, "_fix_scratch", None)  # lives with the engine: launches on one stream reuse it in order
            if fix is None or fix.numel() < n:
                fix = self.e._fix_scratch = torch.zeros(n, dtype=torch.int64, device=self.e.device)
        lib.check(self.L.dcvc_warp_bwd(src.ptr, src.cs, flow.ptr, flow.cs, dout.ptr, dout.cs,
                                       dsrc.ptr if dsrc else None, dsrc.cs if dsrc else 0,
                                       dflow.ptr if dflow else None, dflow.cs if dflow else 0, src.N, src.H, src.W,
                                       src.C, fix.data_ptr() if fix is not None else None, self.stream()), "warp_bwd")

    def _b_up2(self, src, out, scale, out2):
        if self.is_const(src):
            return
        for o in (out, out2):
            if o is None:
                continue
            do = self.grad(o, create=False)
            if do is None:
                continue
            ds = self.grad(src)
            lib.check(self.L.dcvc_up2_bwd(do.ptr, do.cs, ds.ptr, ds.cs, src.N, src.H, src.W, src.C, float(scale),
                                          self.stream()), "up2_bwd")

    def _b_down2(self, src, out, scale):
        do = self.grad(out, create=False)
        if do is None or self.is_const(src):
            return
        ds = self.grad(src)
        lib.check(self.L.dcvc_down2_bwd(do.ptr, do.cs, ds.ptr, ds.cs, src.N, src.H, src.W, src.C, float(scale),
                                        self.stream()), "down2_bwd")

    def _b_maxpool2(self, src, out):
        do = self.grad(out, create=False)
        if do is None or self.is_const(src):
            return
        ds = self.grad(src)
        lib.check(self.L.dcvc_maxpool2_bwd(src.ptr, src.cs, do.ptr, do.cs, ds.ptr, ds.cs, src.N, src.H, src.W, src.C,
                                           self.stream()), "maxpool2_bwd")

    def _b_copy(self, src, out):
        do = self.grad(out, create=False)
        if do is None or self.is_const(src):
            return
        self.accumulate(do, self.grad(src))

    # -- squeeze-excitation ----------------------------------------------------------------
    def _b_se_gate(self, t, w1, w2, mean, gate):
        dg = self.vec.get(gate.data_ptr())
        if dg is None:
            return
        dmean = torch.empty_like(mean)
        dw1 = self.pgrad(w1) if self.wants(w1) else None
        dw2 = self.pgrad(w2) if self.wants(w2) else None
        lib.check(self.L.dcvc_se_bwd(mean.data_ptr(), w1.data_ptr(), w2.data_ptr(), gate.data_ptr(), dg.data_ptr(),
                                     dmean.data_ptr(), dw1.data_ptr() if dw1 is not None else None,
                                     dw2.data_ptr() if dw2 is not None else None, t.N, t.C, w1.shape[0], self.stream()),
                  "se_bwd")
        dt = self.grad(t)
        lib.check(self.L.dcvc_add_channel_vec(dt.ptr, dt.cs, dmean.data_ptr(), 1.0 / t.HW, t.N, t.HW, t.C,
                                              self.stream()), "add_channel_vec")

    # -- quantisation ------------------------------------------------------------------------
    def qstate(self, qkey, q_basic_param, q_scale, N, Cq):
        q = self.q.get(qkey)
        if q is None:
            z = lambda: self.zeros(N * Cq)
            q = dict(param=q_basic_param, q_scale=q_scale, dq_mul=z(), s_div=z(), N=N, C=Cq,
                     dq_scale=self.zeros(N))
            self.q[qkey] = q
        return q

    def _b_scale_channels(self, src, out, q_basic, q_scale, multiply, qkey):
        do = self.grad(out, create=False)
        if do is None:
            return
        ds = self.grad(src)
        lib.check(self.L.dcvc_scale_channels_bwd(do.ptr, do.cs, ds.ptr, ds.cs, q_basic.data_ptr(), q_scale.data_ptr(),
                                                 int(multiply), src.N, src.HW, src.C, self.stream()),
                  "scale_channels_bwd")
        q = self.q[qkey]
        if multiply:
            self.channel_dot(do, src, q["dq_mul"], over_batch=False, accumulate=True)
        else:
            self.channel_dot(do, out, q["s_div"], over_batch=False, accumulate=True)

    def _q_finish(self, q):
        p = q["param"]
        dqb = self.pgrad(p) if self.wants(p) else None
        lib.check(self.L.dcvc_q_finish(q["dq_mul"].data_ptr(), q["s_div"].data_ptr(), p.data_ptr(),
                                       q["q_scale"].data_ptr(), dqb.data_ptr() if dqb is not None else None,
                                       q["dq_scale"].data_ptr(), q["N"], q["C"], self.stream()), "q_finish")

    def _b_round(self, z, z_hat):
        do = self.grad(z_hat, create=False)
        if do is not None:
            self.accumulate(do, self.grad(z))  # straight-through: quant() = x + (round(x) - x).detach()

    def _b_dual_prior(self, step, y, fusion, spatial, params, y_hat, y_res, scales_hat, out, q_basic, q_scale, qkey):
        a = lib.DualPriorBwdArgs()
        a.N, a.H, a.W, a.C, a.step = fusion.N, fusion.H, fusion.W, fusion.C // 3, step
        dres = self.dense.get(y_res.data_ptr()) if y_res is not None else None
        dsh = self.dense.get(scales_hat.data_ptr()) if scales_hat is not None else None
        if dres is not None:
            a.dy_res = dres.data_ptr()
        if dsh is not None:
            a.dscales_hat = dsh.data_ptr()
        if step == 1:
            dsp = self.grad(spatial)
            a.dspatial, a.dspatial_cs = dsp.ptr, dsp.cs
            # remember what step 0 needs from this launch's arguments
            self._dp_out = (out, q_basic, q_scale, qkey)
        else:
            out, q_basic, q_scale, qkey = self._dp_out
            a.y, a.y_cs = y.ptr, y.cs
            a.fusion, a.fusion_cs = fusion.ptr, fusion.cs
            a.y_hat = y_hat.data_ptr()
            do = self.grad(out, create=False)
            if do is not None:
                a.dout, a.dout_cs = do.ptr, do.cs
            dp = self.grad(params, create=False)
            if dp is not None:
                a.dparams, a.dparams_cs = dp.ptr, dp.cs
            dy = self.grad(y)
            a.dy, a.dy_cs = dy.ptr, dy.cs
            df = self.grad(fusion)
            a.dfusion, a.dfusion_cs = df.ptr, df.cs
            dq_plane = torch.empty(a.N * a.H * a.W * a.C, dtype=torch.float32, device=self.e.device)
            a.dq_plane = dq_plane.data_ptr()
            a.q_basic, a.q_scale = q_basic.data_ptr(), q_scale.data_ptr()
        lib.check(self.L.dcvc_dual_prior_bwd(C.byref(a), self.stream()), "dual_prior_bwd")
        if step == 0:
            plane = View(dq_plane.view(a.N, a.H, a.W, a.C), a.C) if a.C % 4 == 0 else None
            assert plane is not None
            self.channel_dot(plane, None, self.q[qkey]["dq_mul"], over_batch=False, accumulate=True)

    # -- rate / distortion ---------------------------------------------------------------------
    def _b_scale_bits(self, name, y_bit, scales_hat, y_res, N, per, kind=0):
        g = self.up.get(name)
        if g is None:
            return
        dy, dsc = torch.empty_like(y_bit), torch.empty_like(scales_hat)
        lib.check(self.L.dcvc_scale_bits_bwd(y_bit.data_ptr(), scales_hat.data_ptr(), g.data_ptr(), dy.data_ptr(),
                                             dsc.data_ptr(), kind, N, per, self.stream()), "scale_bits_bwd")
        self.dense[y_res.data_ptr()] = dy
        self.dense[scales_hat.data_ptr()] = dsc

    def _b_factorized_bits(self, name, z_bit: View, z: View, pblock, params):
        g = self.up.get(name)
        if g is None:
            return
        want = any(self.wants(p) for p in params)
        dblock = self.zeros(pblock.numel()).view(pblock.shape) if want else None
        dz = self.grad(z)
        lib.check(self.L.dcvc_factorized_bits_bwd(z_bit.ptr, z_bit.cs, pblock.data_ptr(), g.data_ptr(), dz.ptr, dz.cs,
                                                  dblock.data_ptr() if want else None, z.N, z.HW, z.C, self.stream()),
                  "factorized_bits_bwd")
        if want:
            # (through the library's own accumulate kernel, not an ATen add: no ATen arithmetic kernel is enqueued while
            # the weight-gradient stream's bf16-MFMA kernels may be running -- tests/test_gpu_backward.py checks the pass)
            C_ = dblock.shape[1]
            for i, p in enumerate(params):
                if self.wants(p):
                    src = View(dblock[i].view(1, 1, 1, C_), C_)
                    dst = View(self.pgrad(p).view(1, 1, 1, C_), C_)
                    self.accumulate(src, dst)

    def _b_sq_err(self, name, a: View, b: View):
        g = self.up.get(name)
        if g is None or self.is_const(a):
            return
        da = self.grad(a)
        lib.check(self.L.dcvc_sq_err_bwd(a.ptr, a.cs, b.ptr, b.cs, g.data_ptr(), da.ptr, da.cs, a.N, a.HW, a.C,
                                         self.stream()), "sq_err_bwd")
