"""``IntraNoAR`` -- the I-frame codec behind the reference's API
(/root/reference/DCVC_HEM/src/models/image_model.py:16-200), on the same HIP kernels.

    IntraNoAR(N=192, anchor_num=4)
    .forward(x, q_scale) -> {x_hat, mse, bit, bpp, bpp_y, bpp_z}           :54-106
    .compress(x, q_scale) -> {"bit_stream"}  (+ "x_hat")                   :148-177
    .decompress(bit_stream, height, width, q_scale) -> {"x_hat"}           :179-200
    .encode_decode(x, q_scale, output_path=None, pic_width=None, pic_height=None)  :114-146
"""
from __future__ import annotations

import time

import torch

from . import entropy as E
from . import lib
from . import stream as S
from .dmc import CodecBase
from .engine import View
from .params import intra_spec


class _IntraFn(torch.autograd.Function):
    """One I picture in training mode as a single autograd node (the counterpart of dmc._FrameFn): forward records a
    grad.Tape on the HIP engine, backward replays it with the kernels of include/dcvc_hip_grad.h.  Inputs after the
    q-scale are the model's parameters; outputs are the per-sample sums (bits_y, bits_z, squared error) and x_hat."""

    @staticmethod
    def forward(ctx, model, x, q, *params):
        from .grad import Tape

        ctx.set_materialize_grads(False)
        tape = Tape(model.engine())
        o, sums = model._train_frame(tape, x.detach(), q.detach())
        ctx.tape, ctx.params, ctx.q_shape, ctx.out_view = tape, params, q.shape, o["x_hat"]
        return sums["bits_y"], sums["bits_z"], sums["sq"], o["x_hat"].nchw()

    @staticmethod
    def backward(ctx, g_y, g_z, g_sq, g_xhat):
        tape = ctx.tape
        if tape is None:
            raise RuntimeError("this picture's tape was already consumed (retain_graph is not supported)")
        e = tape.e
        for name, g in (("bits_y", g_y), ("bits_z", g_z), ("sq", g_sq)):
            if g is not None:
                tape.up[name] = g.detach().to(torch.float32).contiguous()
        if g_xhat is not None:  # a loss computed on the reconstruction by the caller (perceptual terms)
            e.from_nchw(g_xhat, tape.grad(ctx.out_view))
        tape.backward()
        grads = []
        for p, need in zip(ctx.params, ctx.needs_input_grad[3:]):
            g = tape.pgrads.get(id(p)) if need else None
            grads.append(torch.zeros_like(p) if (need and g is None) else g)
        gq = None
        if ctx.needs_input_grad[2]:
            g = tape.q["y"]["dq_scale"]
            n = 1
            for s_ in ctx.q_shape:
                n *= s_
            gq = (g.sum() if n == 1 else g.clone()).reshape(ctx.q_shape)
        ctx.tape = None
        return (None, None, gq, *grads)


class IntraNoAR(CodecBase):
    _tag = "intra"
    _distribution = "gaussian"
    _z_names = ("bit_estimator_z",)

    def __init__(self, N=192, anchor_num=4, seed=0, precision=None):
        super().__init__(intra_spec(N, anchor_num), seed=seed, precision=precision)
        self.N = int(N)
        self.anchor_num = int(anchor_num)

    @staticmethod
    def get_q_scales_from_ckpt(ckpt_path):
        return S.get_state_dict(ckpt_path)["q_scale"].reshape(-1)

    def _synthesis(self, net, y_hat, N, H, W, clamp):
        d = net.decoder_stack("dec", y_hat)
        u = net.unet("refine.0", d)
        k = self._out_set() if self.engine().tape is None else 0  # (a recorded forward owns fresh buffers)
        x_hat = net.buf(f"dpb{k}.x_hat", N=N, H=H, W=W, C=3)
        net.conv("refine.1", u, out=x_hat, out_slope="clamp01" if clamp else None)
        return x_hat

    def get_curr_q(self, q_scale):
        """image_model.py:50-52: max(q_basic, 0.5) * q_scale."""
        return torch.clamp_min(self.P("q_basic"), 0.5) * q_scale

    def _run(self, x, q_scale, mode, tape=None):
        """mode 'estimate' / 'compress' / 'train' (a recorded forward: straight-through rounding, y_res kept)."""
        e = self.engine()
        net = self._net
        Nb, _, H, W = x.shape
        assert H % 64 == 0 and W % 64 == 0, "pad to a multiple of 64 first (stream.get_padding_size)"
        q = self._qvec(q_scale, Nb, "q_scale")
        qb = self.P("q_basic").reshape(-1)
        x3 = self._frame_in("x", x)
        if tape is not None:
            tape.mark_const(x3)  # the picture carries no gradient
        y_raw = net.encoder_stack("enc", x3)
        y = e.scale_channels(y_raw, net.buf("y", like=y_raw, C=self.N), qb, q, qkey="y")
        z = net.hyper_enc5("hyper_enc", y)
        z_hat = net.buf("z_hat", like=z, C=self.N)
        sym_z = e.ibuf("intra/sym_z", Nb * self.N * z.HW) if mode == "compress" else None
        self._wait_coder()
        e.round_symbols(z, z_hat, sym_z)
        fusion = net.three_convs("y_prior_fusion", net.hyper_dec("hyper_dec", z_hat))
        y_hat = net.buf("y_hat", like=y, C=self.N)
        r = self._dual_prior_encode("y", y, fusion, "y_spatial_prior", y_hat, qb, q, want_stats=(mode != "compress"),
                                    want_symbols=(mode == "compress"), want_res=(mode == "train"), qkey="y")
        x_hat = self._synthesis(net, y_hat, Nb, H, W, clamp=(mode == "compress"))  # compress == decoder output
        return dict(N=Nb, H=H, W=W, x3=x3, y=y, z=z, z_hat=z_hat, sym_z=sym_z, r=r, x_hat=x_hat, y_hat=y_hat)

    # ------------------------------------------------------------------ training-mode forward (round 4)
    _noise_override = None  # tests: {"y", "z"} -> NCHW tensors replacing add_noise's draws

    def _noise(self, key, N, H, W, C_):
        """uniform(-0.5, 0.5) like CompressionModel.add_noise (common_model.py:46-49), dense NHWC."""
        if self._noise_override is not None:
            t = self._noise_override[key].to(device=self.device, dtype=torch.float32)
            assert tuple(t.shape) == (N, C_, H, W), (key, t.shape)
            return t.permute(0, 2, 3, 1).contiguous()
        return torch.empty((N, H, W, C_), dtype=torch.float32, device=self.device).uniform_(-0.5, 0.5)

    def _train_frame(self, tape, x, q_scale):
        """Recorded forward of IntraNoAR.forward in training mode (image_model.py:54-100 with self.training: straight-through
        rounding, uniform noise on the residual and on z for the bit estimates, Gaussian likelihood).  The reference's
        trainers run the I-picture codec under no_grad (core/model/dcvc_hem.py:164-167), but its forward IS
        differentiable; this is that path.  Returns the views and the per-sample sums."""
        e = self.engine()
        e.tape = tape
        try:
            e.repack_all()
            N = x.shape[0]
            tape.qstate("y", self.P("q_basic"), self._qvec(q_scale, N, "q_scale"), N, self.N)
            o = self._run(x, tape.q["y"]["q_scale"], "train", tape=tape)
            L, sums = e.L, {}
            sums["sq"] = e.sq_err(o["x_hat"], o["x3"])
            tape.ops.append(("sq_err", "sq", o["x_hat"], o["x3"]))
            lat, r = o["y"], o["r"]
            per = lat.HW * lat.C
            noise = self._noise("y", N, lat.H, lat.W, lat.C)
            y_bit = torch.empty_like(noise).view(-1)
            lib.check(L.dcvc_add_planes(r["y_res"].data_ptr(), lat.C, noise.data_ptr(), lat.C, y_bit.data_ptr(), lat.C,
                                        N * lat.HW, lat.C, e.stream()), "add_planes")
            sums["bits_y"] = e.scale_bits(y_bit, r["scales_hat"], N, per, gaussian=True)
            tape.ops.append(("scale_bits", "bits_y", y_bit, r["scales_hat"], r["y_res"], N, per, 1))
            z = o["z"]
            noise = self._noise("z", N, z.H, z.W, z.C)
            z_bit = View(torch.empty_like(noise), z.C)
            lib.check(L.dcvc_add_planes(z.ptr, z.cs, noise.data_ptr(), z.C, z_bit.ptr, z_bit.cs, N * z.HW, z.C, e.stream()),
                      "add_planes")
            blk = self._zblock("bit_estimator_z")
            sums["bits_z"] = e.factorized_bits(z_bit, blk)
            plist = [self.P(f"bit_estimator_z.f{i}.{k}") for i in (1, 2, 3) for k in ("h", "b", "a")]
            plist += [self.P("bit_estimator_z.f4.h"), self.P("bit_estimator_z.f4.b")]
            tape.ops.append(("factorized_bits", "bits_z", z_bit, z, blk, plist))
            return o, sums
        finally:
            e.tape = None

    def _forward_train(self, x, q_scale):
        q = self.P("q_scale") if q_scale is None else q_scale
        q = q if torch.is_tensor(q) else torch.tensor(float(q), device=self.device)
        params = [p for n, p in self._pmap.items() if n != "q_scale"]
        bits_y, bits_z, sq, x_hat = _IntraFn.apply(self, x, q, *params)
        pix = x.shape[2] * x.shape[3]
        bpp_y, bpp_z = bits_y / pix, bits_z / pix
        return {"x_hat": x_hat, "mse": sq / pix, "bit": (torch.sum(bpp_y + bpp_z) * pix).item(), "bpp": bpp_y + bpp_z,
                "bpp_y": bpp_y, "bpp_z": bpp_z}

    def forward(self, x, q_scale=None):
        """Training mode (round 4): the differentiable forward of image_model.py:54-106 (`bit` stays a float, :102)."""
        if self.training:
            return self._forward_train(x, q_scale)
        with torch.no_grad():
            return self._forward_eval(x, q_scale)

    def _forward_eval(self, x, q_scale=None):
        e = self.engine()
        o = self._run(x, q_scale, "estimate")
        pix = o["H"] * o["W"]
        bpp_y = e.scale_bits(o["r"]["y_q"], o["r"]["scales_hat"], o["N"], o["y"].HW * self.N, gaussian=True) / pix
        bpp_z = e.factorized_bits(o["z_hat"], self._zblock("bit_estimator_z")) / pix
        mse = e.sq_err(o["x3"], o["x_hat"]) / pix
        return {"x_hat": o["x_hat"].nchw(), "mse": mse, "bit": (torch.sum(bpp_y + bpp_z) * pix).item(),
                "bpp": bpp_y + bpp_z, "bpp_y": bpp_y, "bpp_z": bpp_z, "_views": o}

    @torch.no_grad()
    def compress(self, x, q_scale, defer=False, coder="host", check_range=True):
        if self.entropy_coder is None:
            raise RuntimeError("call update() before compress()/decompress()")
        o = self._run(x, q_scale, "compress")
        N = o["N"]  # N > 1: a batch of rate points (one q-scale per element), one independent stream each
        assert coder in ("host", "device")
        zs = o["z_hat"]
        pending = (self._stage_symbols if coder == "host" else self._stage_symbols_device)([  # image_model.py:168-171
            ("bit_estimator_z", o["sym_z"], None, (N, self.N, zs.H, zs.W)),
            ("scale", o["r"]["sym"][0], o["r"]["idx"][0], None),
            ("scale", o["r"]["sym"][1], o["r"]["idx"][1], None),
        ], batch=N)
        if defer:
            return {"pending": pending, "x_hat": o["x_hat"].nchw(), "_views": o}
        streams = pending.finish_all()
        if check_range:  # split-fp16 range guard (DMC.compress says when a caller reads it itself)
            self.engine().check_status()
        return {"bit_stream": streams[0], "bit_streams": streams, "x_hat": o["x_hat"].nchw(), "_views": o}

    @torch.no_grad()
    def decompress(self, bit_stream, height, width, q_scale, coder=None, defer_check=False, check_range=True):
        self._defer_check = defer_check
        if self.entropy_coder is None:
            raise RuntimeError("call update() before compress()/decompress()")
        if coder is None:
            coder = "device" if bit_stream[:4] == E.DRANS_MAGIC else "host"
        self._dc_active = coder == "device"
        try:
            r = self._decompress(bit_stream, height, width, q_scale)
        finally:
            self._dc_active = False
        if check_range:
            self.engine().check_status()
        return r

    def _decompress(self, bit_stream, height, width, q_scale):
        e = self.engine()
        net = self._net
        q = self._qvec(q_scale, 1, "q_scale")
        if self._dc_active:
            self.device_coder().set_stream(bit_stream)
        else:
            self.entropy_coder.set_stream(bit_stream)
        zh, zw = S.get_downsampled_shape(height, width, 64)
        sym = self._decode_factorized("bit_estimator_z", 1, self.N, zh, zw)
        z_hat = e.symbols_to_nhwc(sym, net.buf("z_hat", N=1, H=zh, W=zw, C=self.N))
        fusion = net.three_convs("y_prior_fusion", net.hyper_dec("hyper_dec", z_hat))
        y_hat = net.buf("y_hat", N=1, H=zh * 4, W=zw * 4, C=self.N)
        self._dual_prior_decode("y", fusion, "y_spatial_prior", y_hat, self.P("q_basic").reshape(-1), q)
        x_hat = self._synthesis(net, y_hat, 1, zh * 64, zw * 64, clamp=self._clamp_decoded)  # .clamp_(0, 1), :199
        if self._dc_active:
            self._dcoder.release()
            if not self._defer_check:
                self._dcoder.check()
        return {"x_hat": x_hat.nchw()}

    def encode_decode(self, x, q_scale, output_path=None, pic_width=None, pic_height=None):
        if output_path is None:
            return self.forward(x, q_scale)
        assert pic_height is not None and pic_width is not None
        t0 = time.time()
        q_scale, q_index = S.get_rounded_q(q_scale)
        compressed = self.compress(x, q_scale)
        S.encode_i(pic_height, pic_width, q_index, compressed["bit_stream"], output_path)
        bit = S.filesize(output_path) * 8
        t1 = time.time()
        height, width, q_index, bit_stream = S.decode_i(output_path)
        decompressed = self.decompress(bit_stream, height, width, q_index / 100, coder="host")
        torch.cuda.synchronize(self.device)
        t2 = time.time()
        return {"bit": bit, "x_hat": decompressed["x_hat"], "encoding_time": t1 - t0, "decoding_time": t2 - t1}
