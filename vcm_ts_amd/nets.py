"""The codec networks as launch programs over the HIP engine.

Each function enqueues the kernels of one reference sub-network on strided-NHWC views and
returns the view(s) holding its result.  Layers are addressed by the reference's
state-dict names (vcm_ts_amd/params.py), the structure follows

  /root/reference/DCVC_HEM/src/models/video_net.py   (ResBlock :74, MEBasic :99, ME_Spynet :118,
                                                      SELayer :149, ConvBlockResidual :165, UNet :182,
                                                      get_enc_dec_models :226, get_hyper_enc_dec_models :251)
  /root/reference/DCVC_HEM/src/layers/layers.py      (ResidualBlock* :42-127, subpel_conv* :23-34)
  /root/reference/DCVC_HEM/src/models/video_model.py (FeatureExtractor :17, MultiScaleContextFusion :40,
                                                      ContextualEncoder :71, ContextualDecoder :93,
                                                      ReconGeneration :115, motion_compensation :233)

but not its execution: activations, residual adds, SE gates, PixelShuffle and torch.cat
are folded into the convolution launches (engine.conv), and concatenations are either
multi-segment convolution inputs or channel slices of one wider buffer.
"""
from __future__ import annotations

from .engine import Engine, View


class Net:
    def __init__(self, engine: Engine, params, tag: str):
        self.e = engine
        self.p = params  # callable: name -> tensor
        self.tag = tag   # workspace namespace ("dmc" / "intra")

    # ------------------------------------------------------------------ helpers
    def buf(self, name, like: View = None, C=None, N=None, H=None, W=None, cs=None, zero=False) -> View:
        if like is not None:
            N, H, W = like.N, like.H, like.W
        return self.e.buf(f"{self.tag}/{name}", N, H, W, C, cs=cs, zero=zero)

    def conv(self, name, srcs, out: View = None, stride=1, in_slope=None, out_slope=None, ps=False, res=None,
             gate=None, res2=None, out_name=None, cin_slice=None, want_chan_sums=False):
        """want_chan_sums: also return (buffer, rows, row stride) of the output's per-workgroup channel sums
        (fused SE squeeze; not while a training tape records, which keeps the separate reduction)."""
        if isinstance(srcs, View):
            srcs = [srcs]
        w = self.p(name + ".weight")
        seg_C = tuple(s.C for s in srcs)
        pk = self.e.pack((self.tag, name, seg_C, ps, cin_slice), w, self.p(name + ".bias"), seg_C, ps, cin_slice)
        if out is None:
            s0 = srcs[0]
            pad = pk.ks // 2
            Ho = (s0.H + 2 * pad - pk.ks) // stride + 1
            Wo = (s0.W + 2 * pad - pk.ks) // stride + 1
            m = 2 if ps else 1
            out = self.buf(out_name or name, N=s0.N, H=Ho * m, W=Wo * m, C=pk.Cout // 4 if ps else pk.Cout)
        if want_chan_sums:
            partial = None
            if self.e.tape is None and not ps and out.cs % 4 == 0:
                buf, parts = self.e.chan_partial_buf(f"{self.tag}/{name}", pk, out, stride)
                partial = (buf, parts, pk.Cout_pad)
            v = self.e.conv(pk, srcs, out, stride=stride, in_slope=in_slope, out_slope=out_slope, res=res, gate=gate,
                            res2=res2, chan_partial=None if partial is None else partial[0])
            return v, partial
        return self.e.conv(pk, srcs, out, stride=stride, in_slope=in_slope, out_slope=out_slope, res=res, gate=gate,
                           res2=res2)

    # ------------------------------------------------------------------ blocks
    def res_block(self, name, x: View, slope=0.01, end_with_relu=False, out: View = None, res2: View = None) -> View:
        """video_net.ResBlock: x + [act](conv2(act(conv1(act(x)))))."""
        a = self.conv(name + ".conv1", x, in_slope=slope, out_slope=slope)
        return self.conv(name + ".conv2", a, out=out, out_slope=slope if end_with_relu else None, res=x, res2=res2)

    def residual_block(self, name, x: View, out: View = None) -> View:
        """layers.ResidualBlock."""
        a = self.conv(name + ".conv1", x, out_slope=0.01)
        return self.conv(name + ".conv2", a, out=out, out_slope=0.01, res=x)

    def residual_block_stride(self, name, x: View) -> View:
        """layers.ResidualBlockWithStride."""
        a = self.conv(name + ".conv1", x, stride=2, out_slope=0.01)
        idn = self.conv(name + ".downsample", x, stride=2)
        return self.conv(name + ".conv2", a, out_slope=0.1, res=idn)

    def residual_block_up(self, name, x: View) -> View:
        """layers.ResidualBlockUpsample."""
        idn = self.conv(name + ".upsample.0", x, ps=True)
        a = self.conv(name + ".subpel_conv.0", x, ps=True, out_slope=0.01)
        return self.conv(name + ".conv", a, out_slope=0.1, res=idn)

    def encoder_stack(self, name, x: View) -> View:
        for i in (0, 2, 4):
            x = self.residual_block_stride(f"{name}.{i}", x)
            x = self.residual_block(f"{name}.{i + 1}", x)
        return self.conv(f"{name}.6", x, stride=2)

    def decoder_stack(self, name, x: View, out: View = None) -> View:
        for i in (0, 2, 4):
            x = self.residual_block(f"{name}.{i}", x)
            x = self.residual_block_up(f"{name}.{i + 1}", x)
        x = self.residual_block(f"{name}.6", x)
        return self.conv(f"{name}.7.0", x, ps=True, out=out)

    def hyper_enc5(self, name, x: View) -> View:
        for i, s in ((0, 1), (2, 1), (4, 2), (6, 1)):
            x = self.conv(f"{name}.{i}", x, stride=s, out_slope=0.01)
        return self.conv(f"{name}.8", x, stride=2)

    def hyper_dec(self, name, x: View, out: View = None) -> View:
        x = self.conv(f"{name}.0", x, out_slope=0.01)
        x = self.conv(f"{name}.2.0", x, ps=True, out_slope=0.01)
        x = self.conv(f"{name}.4", x, out_slope=0.01)
        x = self.conv(f"{name}.6.0", x, ps=True, out_slope=0.01)
        return self.conv(f"{name}.8", x, out=out)

    def three_convs(self, name, srcs, slope=0.2, cin_slice=None) -> View:
        x = self.conv(f"{name}.0", srcs, out_slope=slope, cin_slice=cin_slice)
        x = self.conv(f"{name}.2", x, out_slope=slope)
        return self.conv(f"{name}.4", x)

    def se_block(self, name, srcs, out: View = None) -> View:
        """ConvBlockResidual: up_dim(x) + conv.2(leaky(conv.0(x))) * SE gate."""
        a = self.conv(f"{name}.conv.0", srcs, out_slope=0.01)
        t, sums = self.conv(f"{name}.conv.2", a, want_chan_sums=True)  # SE squeeze rides on this launch's epilogue
        gate = self.e.se_gate(f"{self.tag}/{name}", t, self.p(f"{name}.conv.3.fc.0.weight"),
                              self.p(f"{name}.conv.3.fc.2.weight"), partial=sums)
        return self.conv(f"{name}.up_dim", srcs, out=out, res=t, gate=gate)

    def unet(self, name, x: View, out: View = None) -> View:
        x1 = self.se_block(f"{name}.conv1", x)
        p1 = self.e.maxpool2(x1, self.buf(f"{name}.pool1", N=x.N, H=x.H // 2, W=x.W // 2, C=x1.C))
        x2 = self.se_block(f"{name}.conv2", p1)
        p2 = self.e.maxpool2(x2, self.buf(f"{name}.pool2", N=x.N, H=x.H // 4, W=x.W // 4, C=x2.C))
        x3 = self.se_block(f"{name}.conv3", p2)
        for i in range(4):
            x3 = self.res_block(f"{name}.context_refine.{i}", x3, slope=0.0)
        u3 = self.conv(f"{name}.up3.0", x3, ps=True)
        d3 = self.se_block(f"{name}.up_conv3", [x2, u3])
        u2 = self.conv(f"{name}.up2.0", d3, ps=True)
        return self.se_block(f"{name}.up_conv2", [x1, u2], out=out)

    # ------------------------------------------------------------------ SpyNet
    def spynet(self, x8: View, ref: View) -> View:
        """ME_Spynet.  x8: the current frame in channels 0-2 of the finest level's 8-channel
        MEBasic input buffer; ref: reference frame (3 channels).  Returns the flow (2 ch)."""
        N, H, W = x8.N, x8.H, x8.W
        in_views = [View(x8.base, 8)]
        refs = [ref]
        for k in range(1, 4):
            h, w = H >> k, W >> k
            v = self.buf(f"spy.in{k}", N=N, H=h, W=w, C=8)
            self.e.down2(in_views[k - 1].slice(0, 3), v.slice(0, 3), avgpool_order=True)
            in_views.append(v)
            refs.append(self.e.down2(refs[k - 1], self.buf(f"spy.ref{k}", N=N, H=h, W=w, C=3), avgpool_order=True))
        flow = self.buf("spy.flow_init", N=N, H=H >> 4, W=W >> 4, C=2, zero=True)
        for lvl in range(4):
            k = 3 - lvl
            v = in_views[k]
            fu = self.e.up2(flow, v.slice(6, 2), scale=2.0)
            self.e.warp(refs[k], fu, v.slice(3, 3))
            base = f"optic_flow.moduleBasic.{lvl}"
            t = v
            for i in (1, 2, 3, 4):
                t = self.conv(f"{base}.conv{i}", t, out_slope=0.0)
            flow = self.conv(f"{base}.conv5", t, res=fu)
        return flow

    # ------------------------------------------------------------------ P-frame pieces
    def feature_pyramid(self, ref_frame: View, ref_feature):
        """feature_adaptor_I/P + FeatureExtractor (video_model.py:17-38,226-232): depends only on the
        DPB, not on the motion of the current picture, so the codec may run it on a side stream
        while SpyNet and the motion-vector codec run (DMC._run)."""
        if ref_feature is None:
            f = self.conv("feature_adaptor_I", ref_frame)
        else:
            f = self.conv("feature_adaptor_P", ref_feature)
        l1 = self.res_block("feature_extractor.res_block1", self.conv("feature_extractor.conv1", f))
        l2 = self.res_block("feature_extractor.res_block2", self.conv("feature_extractor.conv2", l1, stride=2))
        l3 = self.res_block("feature_extractor.res_block3", self.conv("feature_extractor.conv3", l2, stride=2))
        return l1, l2, l3

    def motion_compensation(self, ref_frame: View, ref_feature, mv: View, enc_cat2: View, enc_cat3: View,
                            want_warp_frame: bool, pyramid=None):
        """video_model.py:226-246 + MultiScaleContextFusion :40-68.  context2/context3 are written
        straight into the second halves of the contextual encoder's concat buffers."""
        N, H, W = mv.N, mv.H, mv.W
        warp_frame = None
        if want_warp_frame:
            warp_frame = self.e.warp(ref_frame, mv, self.buf("warp_frame", like=mv, C=3))
        mv2 = self.e.down2(mv, self.buf("mv2", N=N, H=H // 2, W=W // 2, C=2), scale=0.5)
        mv3 = self.e.down2(mv2, self.buf("mv3", N=N, H=H // 4, W=W // 4, C=2), scale=0.5)
        l1, l2, l3 = pyramid if pyramid is not None else self.feature_pyramid(ref_frame, ref_feature)
        cat1 = self.buf("fusion_cat1", like=l1, C=128)  # [context2_up | warped context1]
        c1w = self.e.warp(l1, mv, cat1.slice(64, 64))
        c3w = self.e.warp(l3, mv3, self.buf("ctx3_warp", like=l3, C=64))
        n = "context_fusion_net"
        cat2 = self.buf("fusion_cat2", like=l2, C=128)  # [context3_up | warped context2]
        c2w = self.e.warp(l2, mv2, cat2.slice(64, 64))
        self.res_block(f"{n}.res_block3_up", self.conv(f"{n}.conv3_up.0", c3w, ps=True), out=cat2.slice(0, 64))
        c3 = self.res_block(f"{n}.res_block3_out", self.conv(f"{n}.conv3_out", c3w), out=enc_cat3.slice(64, 64),
                            res2=c3w)
        self.res_block(f"{n}.res_block2_up", self.conv(f"{n}.conv2_up.0", cat2, ps=True), out=cat1.slice(0, 64))
        c2 = self.res_block(f"{n}.res_block2_out", self.conv(f"{n}.conv2_out", cat2), out=enc_cat2.slice(64, 64),
                            res2=c2w)
        c1 = self.res_block(f"{n}.res_block1_out", self.conv(f"{n}.conv1_out", cat1), out=self.buf("context1", like=l1, C=64),
                            res2=c1w)
        return c1, c2, c3, warp_frame

    def contextual_encoder(self, x3: View, c1: View, enc_cat2: View, enc_cat3: View) -> View:
        """video_model.py:71-90.  enc_cat2 = [feature | context2], enc_cat3 = [feature | context3]."""
        n = "contextual_encoder"
        self.conv(f"{n}.conv1", [x3, c1], stride=2, out=enc_cat2.slice(0, 64))
        f = self.res_block(f"{n}.res1", enc_cat2, slope=0.1, end_with_relu=True)
        self.conv(f"{n}.conv2", f, stride=2, out=enc_cat3.slice(0, 64))
        f = self.res_block(f"{n}.res2", enc_cat3, slope=0.1, end_with_relu=True)
        return self.conv(f"{n}.conv4", self.conv(f"{n}.conv3", f, stride=2), stride=2)

    def _beside_context(self, name, ctx: View) -> View:
        """The 128-channel buffer [64 new channels | ctx] the contextual decoder's ResBlocks run on (torch.cat((f, context),
        video_model.py:103,107).  Inference: motion_compensation wrote the context into channels 64..127 of the ENCODER's
        concat buffer, whose first half (the contextual encoder's features) is dead once y exists -- and in a decode-only
        call was never written -- so the decoder's features go there and no copy of the context is made (round 4: two
        copy launches per picture less, same bits).  A recorded training forward keeps every buffer: it copies."""
        if (self.e.tape is None and ctx.cs == 128 and ctx.coff == 64 and ctx.base.dim() == 4
                and tuple(ctx.base.shape) == (ctx.N, ctx.H, ctx.W, 128)):
            return View(ctx.base, 128)
        cat = self.buf(name, like=ctx, C=128)
        self.e.copy(ctx, cat.slice(64, 64))
        return cat

    def contextual_decoder(self, y_hat: View, c2: View, c3: View) -> View:
        """video_model.py:93-112."""
        n = "contextual_decoder"
        u1 = self.conv(f"{n}.up1.0", y_hat, ps=True)
        cat3 = self._beside_context("dec_cat3", c3)
        self.conv(f"{n}.up2.0", u1, ps=True, out=cat3.slice(0, 64))
        f = self.res_block(f"{n}.res1", cat3, slope=0.1, end_with_relu=True)
        cat2 = self._beside_context("dec_cat2", c2)
        self.conv(f"{n}.up3.0", f, ps=True, out=cat2.slice(0, 64))
        f = self.res_block(f"{n}.res2", cat2, slope=0.1, end_with_relu=True)
        return self.conv(f"{n}.up4.0", f, ps=True)

    def recon_generation(self, dec_feature: View, c1: View, feature_out: View, recon_out: View, clamp=False):
        """video_model.py:115-128 (called as (recon_image_feature, context1))."""
        n = "recon_generation_net"
        f = self.conv(f"{n}.first_conv", [dec_feature, c1])
        f = self.unet(f"{n}.unet_1", f)
        f = self.unet(f"{n}.unet_2", f, out=feature_out)
        self.conv(f"{n}.recon_conv", f, out=recon_out, out_slope="clamp01" if clamp else None)
        return feature_out, recon_out
