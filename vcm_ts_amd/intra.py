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
from . import stream as S
from .dmc import CodecBase
from .params import intra_spec


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
        k = self._out_set()
        x_hat = net.buf(f"dpb{k}.x_hat", N=N, H=H, W=W, C=3)
        net.conv("refine.1", u, out=x_hat, out_slope="clamp01" if clamp else None)
        return x_hat

    def get_curr_q(self, q_scale):
        """image_model.py:50-52: max(q_basic, 0.5) * q_scale."""
        return torch.clamp_min(self.P("q_basic"), 0.5) * q_scale

    def _run(self, x, q_scale, mode):
        e = self.engine()
        net = self._net
        Nb, _, H, W = x.shape
        assert H % 64 == 0 and W % 64 == 0, "pad to a multiple of 64 first (stream.get_padding_size)"
        q = self._qvec(q_scale, Nb, "q_scale")
        qb = self.P("q_basic").reshape(-1)
        x3 = self._frame_in("x", x)
        y_raw = net.encoder_stack("enc", x3)
        y = e.scale_channels(y_raw, net.buf("y", like=y_raw, C=self.N), qb, q)
        z = net.hyper_enc5("hyper_enc", y)
        z_hat = net.buf("z_hat", like=z, C=self.N)
        sym_z = e.ibuf("intra/sym_z", Nb * self.N * z.HW) if mode == "compress" else None
        self._wait_coder()
        e.round_symbols(z, z_hat, sym_z)
        fusion = net.three_convs("y_prior_fusion", net.hyper_dec("hyper_dec", z_hat))
        y_hat = net.buf("y_hat", like=y, C=self.N)
        r = self._dual_prior_encode("y", y, fusion, "y_spatial_prior", y_hat, qb, q, want_stats=(mode == "estimate"),
                                    want_symbols=(mode == "compress"))
        x_hat = self._synthesis(net, y_hat, Nb, H, W, clamp=(mode == "compress"))  # compress == decoder output
        return dict(N=Nb, H=H, W=W, x3=x3, y=y, z_hat=z_hat, sym_z=sym_z, r=r, x_hat=x_hat, y_hat=y_hat)

    @torch.no_grad()
    def forward(self, x, q_scale=None):
        self._eval_only()
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
