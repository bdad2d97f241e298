/* dcvc_hip.h -- C ABI of the MI355X (gfx950) kernel layer for the DCVC-HEM per-frame path.
 *
 * Every entry point takes raw device pointers, plain sizes and a hipStream_t passed as
 * void*; it returns 0 on success or a negative DCVC_E_* code (no exceptions, no torch
 * types).  Work is enqueued on the given stream and is stream-ordered; buffers are owned by
 * the caller.
 *
 * Activation layout in HBM: fp32 "NHWC with channel stride": element (n, y, x, c) of a
 * tensor lives at  base[((n*H + y)*W + x)*cs + c],  cs >= C, cs % 4 == 0, base 16-byte
 * aligned.  A channel slice of a wider buffer (zero-copy concatenation) is the same thing
 * with base advanced by the channel offset.
 *
 * What each entry point replaces in the reference (file:line under /root/reference):
 *   dcvc_conv2d             nn.Conv2d + bias + (Leaky)ReLU + residual add + nn.PixelShuffle(2)
 *                           + torch.cat prologue: DCVC_HEM/src/layers/layers.py:18-127,
 *                           src/models/video_net.py:74-115,165-223, video_model.py:17-128
 *   dcvc_conv_pack_weights  (host) weight re-layout for dcvc_conv2d, run once per layer
 *   dcvc_warp               flow_warp/torch_warp (F.grid_sample bilinear/border/align_corners)
 *                           src/models/video_net.py:32-55
 *   dcvc_up2                bilinearupsacling(flow) * 2.0   video_net.py:58-63,139
 *   dcvc_down2              bilineardownsacling(x) * scale  video_net.py:66-71, video_model.py:237-238
 *                           and F.avg_pool2d(x, 2, 2)       video_net.py:132-133
 *   dcvc_maxpool2           nn.MaxPool2d(2)                 video_net.py:185
 *   dcvc_channel_mean, dcvc_se_gate   SELayer               video_net.py:149-162
 *   dcvc_nchw_to_nhwc / dcvc_nhwc_to_nchw   boundary layout change (reference tensors are NCHW)
 *   dcvc_scale_channels     y / curr_q, y_hat * curr_q      video_model.py:485,502,511,531
 *   dcvc_round_symbols      torch.round(z) + .int()         video_model.py:284,310; entropy_models.py:185
 *   dcvc_dual_prior_enc     CompressionModel.forward_dual_prior / process_with_mask / get_mask
 *                           + GaussianEncoder.build_indexes src/models/common_model.py:82-177,
 *                           src/entropy_models/entropy_models.py:264-268
 *   dcvc_dual_prior_dec_*   CompressionModel.decompress_dual_prior  common_model.py:182-217
 *   dcvc_scale_bits (kind 0 Laplace / 1 Gaussian) / dcvc_factorized_bits / dcvc_sq_err
 *                           get_y_laplace_bits, get_y_gaussian_bits, get_z_bits, probs_to_bits,
 *                           nn.MSELoss + per-sample sums    common_model.py:51-73, video_model.py:538-571
 *   dcvc_symbols_to_nhwc    decoded symbols back to a float tensor   entropy_models.py:189-195,276-281
 *   dcvc_copy_channels      the materialised halves of torch.cat where a residual needs them
 */
#ifndef DCVC_HIP_H
#define DCVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCVC_OK 0
#define DCVC_E_ARG (-1)      /* bad argument (shape, alignment, unsupported kernel size) */
#define DCVC_E_LAUNCH (-2)   /* hipGetLastError() after the launch was not hipSuccess */
#define DCVC_E_RANGE (-3)    /* a weight does not fit the split-fp16 representation (|w| >= 1023.5): the packed
                                buffer holds the clamped value; use DCVC_PREC_FP32 for this layer */

/* bits of the optional device status word of the split-fp16 kernels */
#define DCVC_STATUS_ACT_SATURATED 1  /* an activation with |v| > 8188 was clamped while being stored split */

#define DCVC_MAX_SEG 3

/* Arithmetic of dcvc_conv2d.  FP32: v_mfma_f32_32x32x2_f32, bitwise an fp32 fmaf chain ("parity
 * mode").  FP16X3: every fp32 operand split into fp16 hi + lo, three v_mfma_f32_32x32x16_f16 per
 * product with fp32 accumulation; operand representation error <= max(2^-22 |v|, ~4e-9),
 * dropped term <= 2^-22 |x w|; |activation| must stay below 8188 and |weight| below 1023
 * (clamped).  Weights must be packed for the precision they are used with. */
#define DCVC_PREC_FP32 0
#define DCVC_PREC_FP16X3 1

/* One input segment of a convolution: the kernel walks segments in order, which is the
 * channel order of the torch.cat the reference would have materialised. */
typedef struct {
    const float *ptr;   /* (N, H, W, cs) */
    int32_t C;          /* channels used from this segment */
    int32_t cs;         /* channel stride in floats */
} dcvc_seg;

typedef struct {
    /* input */
    dcvc_seg seg[DCVC_MAX_SEG];
    int32_t nseg;
    int32_t N, Hin, Win;
    int32_t in_act;       /* 0: none, 1: LeakyReLU(in_slope) applied to the input on load */
    float in_slope;
    /* filter: packed by dcvc_conv_pack_weights */
    const float *wpack;
    const float *bpack;
    int32_t ks;           /* 1, 3 or 7 (padding ks/2) */
    int32_t stride;       /* 1 or 2 */
    int32_t Cout;         /* real output channels (before pixel shuffle) */
    int32_t Cout_pad;     /* as returned by dcvc_conv_pack_weights */
    /* output */
    float *out;           /* (N, Hout*ps, Wout*ps, out_cs) where ps = pixel_shuffle ? 2 : 1 */
    int32_t out_cs;
    int32_t out_act;      /* 0: none, 1: LeakyReLU(out_slope) after bias, 2: clamp to [0, 1], 3 (dcvc_conv2d only;
                             round 4): res2 is not added but used as a mask source, out = out * (res2 > 0 ? 1 : out_slope) [+ res] --
                             the backward of a producer's LeakyReLU applied by the data-gradient convolution of its only
                             consumer, so that the producer needs no epilogue-backward pass of its own (grad.py) */
    float out_slope;
    int32_t pixel_shuffle;/* 1: write PixelShuffle(2) of the result (Cout % 4 == 0) */
    const float *res;     /* optional residual added after the activation, laid out like out */
    int32_t res_cs;
    const float *res_gate;/* optional (N, Cfinal) per-sample per-channel factor on res (SE gate) */
    const float *res2;    /* optional second residual, added last: out = res2 + (act(conv) + res) */
    int32_t res2_cs;
    int32_t precision;    /* DCVC_PREC_* ; must match the packing of wpack */
    int32_t *status;      /* optional device word: OR-ed with DCVC_STATUS_ACT_SATURATED when an OUTPUT of this launch
                             has |v| > 8188, the magnitude a DCVC_PREC_FP16X3 consumer would clamp on load.  NULL: no
                             check (and no cost).  Weights are range-checked at pack time (DCVC_E_RANGE). */
    float *chan_partial;  /* optional (N, parts, Cout_pad) floats, parts = dcvc_conv_chan_partial_parts(): every
                             workgroup writes the per-channel sums of the outputs it stored, so that SELayer's global
                             average pool (video_net.py:149-162) costs no second pass over the tensor; finish with
                             dcvc_channel_mean_finish.  Not with pixel_shuffle; needs the 16-byte aligned epilogue. */
    int32_t tile_row0;    /* with tile_rows > 0: compute only the output rows of tile rows [tile_row0, tile_row0 +    */
    int32_t tile_rows;    /* tile_rows) (a tile row = dcvc_conv_tile_rows() output rows; pixels outside the band are   */
                          /* left untouched, inputs above / below it are read as usual).  0, 0: the whole picture.     */
                          /* A host that runs consecutive layers band by band keeps a layer's output in the 256 MB    */
                          /* Infinity Cache until its consumer reads it (vcm_ts_amd/engine.py: banded launches).       */
    int32_t pair_taps;    /* 1: wpack comes from dcvc_conv_pack_weights_paired (7x7, stride 1, ONE input segment of    */
                          /* <= 8 channels, DCVC_PREC_FP16X3: SpyNet's first layer, flow_estimation.py MEBasic conv1): */
                          /* two taps share a 16-deep K step instead of padding 8 channels to 16.  dcvc_conv2d only.  */
} dcvc_conv_args;

/* output rows per tile row of dcvc_conv2d / dcvc_conv2d_k32 for this kernel size and stride (the band granularity) */
int32_t dcvc_conv_tile_rows(int32_t ks, int32_t stride);

/* number of partial rows per image dcvc_conv2d writes to chan_partial for this output size */
int32_t dcvc_conv_chan_partial_parts(int32_t ks, int32_t stride, int32_t Hout, int32_t Wout);
/* mean(n, c) = sum over the partial rows (fixed order) / HW */
int dcvc_channel_mean_finish(const float *chan_partial, int32_t parts, int32_t row_stride, float *mean, int32_t N,
                             int32_t C, int32_t HW, void *stream);

/* Number of floats dcvc_conv_pack_weights writes to wpack for this geometry, and the padded
 * output-channel count (bias length) through *cout_pad.  seg_C: channels per segment. */
int64_t dcvc_conv_pack_size(int32_t Cout, int32_t ks, int32_t nseg, const int32_t *seg_C, int32_t *cout_pad);

/* HOST function.  w: (Cout, sum(seg_C), ks, ks) fp32 as nn.Conv2d stores it; b: (Cout) or
 * NULL.  Writes host buffers wpack / bpack which the caller uploads once. */
int dcvc_conv_pack_weights(const float *w, const float *b, int32_t Cout, int32_t ks, int32_t nseg,
                           const int32_t *seg_C, int32_t pixel_shuffle, int32_t precision, float *wpack,
                           float *bpack);

/* Tap-paired packing for dcvc_conv_args.pair_taps (7x7 filters, Cin <= 8, split fp16).  w is (Cout, Cin, 7, 7) as
 * nn.Conv2d stores it; returns DCVC_E_RANGE when a |weight| >= 1023.5 had to be clamped. */
int64_t dcvc_conv_pack_size_paired(int32_t Cout, int32_t Cin, int32_t *cout_pad);
int dcvc_conv_pack_weights_paired(const float *w, const float *b, int32_t Cout, int32_t Cin, float *wpack, float *bpack);

int dcvc_conv2d(const dcvc_conv_args *a, void *stream);

/* ---- layers with at most 16 output channels (DCVC_PREC_FP16X3 only) -------------------------------------
 * SpyNet's 32->16 and 16->2 7x7 layers, the 64->3 reconstruction layer (video_net.py:99-115,
 * video_model.py:115-128): 16-pixel x 16-channel tiles on v_mfma_f32_16x16x32_f16 whose 32-deep K carries the
 * hi/lo operand split (vcm_ts_amd/csrc/conv_small.hip), instead of padding the channels to 32.  Same
 * dcvc_conv_args, restricted to ks 3 or 7, stride 1, no pixel shuffle / gate / res2 / chan_partial; weights
 * from dcvc_conv_small_pack_weights (HOST; DCVC_E_RANGE as dcvc_conv_k32_pack_weights). */
int64_t dcvc_conv_small_pack_bytes(int32_t Cout, int32_t ks, int32_t nseg, const int32_t *seg_C);
int dcvc_conv_small_pack_weights(const float *w, const float *b, int32_t Cout, int32_t ks, int32_t nseg,
                                 const int32_t *seg_C, void *wpack, float *bpack);
int dcvc_conv2d_small(const dcvc_conv_args *a, void *stream);

/* ---- 32-channel-chunk form of DCVC_PREC_FP16X3 (round 3) ------------------------------------------------------
 * The same operator as dcvc_conv2d(DCVC_PREC_FP16X3) for stride-1 1x1 / 3x3 layers whose input segments are all
 * multiples of 32 channels, on v_mfma_f32_16x16x32_f16 (vcm_ts_amd/csrc/conv_k32.hip): one tap of a 32-channel chunk
 * is one K step.  Measured on MI355X the 16x16x32 shape sustains 1.19x the FLOP/s of 32x32x16 in the bare
 * fragment-read + MFMA loop (profiles/r03_mfma_loop_probe.txt).  Same argument struct; precision must be
 * DCVC_PREC_FP16X3, stride 1, 16-byte-aligned epilogue ((Cout or Cout/4 with pixel shuffle) % 4 == 0 and aligned
 * out / res / res2); `status` is cheap here (two VALU per four outputs) and meant to be always passed.  Weights from
 * dcvc_conv_k32_pack_weights (HOST; returns DCVC_E_RANGE with the buffers written, clamped, when a |weight| >= 1023.5).
 * Deterministic, but not bit-identical to dcvc_conv2d: the instruction sums 32 products per step instead of 16.
 * Limits (DCVC_E_ARG beyond them; callers fall back to dcvc_conv2d): an image of any operand below 4 GiB, fewer
 * than 2^24 output pixels per image (after pixel shuffle), channel strides below 2^22 floats. */
int64_t dcvc_conv_k32_pack_bytes(int32_t Cout, int32_t ks, int32_t nseg, const int32_t *seg_C, int32_t *cout_pad);
int dcvc_conv_k32_pack_weights(const float *w, const float *b, int32_t Cout, int32_t ks, int32_t nseg,
                               const int32_t *seg_C, int32_t pixel_shuffle, void *wpack, float *bpack);
int dcvc_conv2d_k32(const dcvc_conv_args *a, void *stream);
/* Developer A/B hook: waves per workgroup (8, the default, or 4) of the 64-output-channel 3x3 kernel.  Results are
 * bit-identical either way (sums inside a tile are ordered by tile row, not by wave); process-wide, not thread-safe. */
int dcvc_conv_k32_set_waves(int32_t waves);

/* ---- resampling ------------------------------------------------------------------------ */
/* out(n,y,x,c) = bilinear(src(n,.,.,c), x + flow(n,y,x,0), y + flow(n,y,x,1)), border clamp */
int dcvc_warp(const float *src, int32_t src_cs, const float *flow, int32_t flow_cs, float *out, int32_t out_cs,
              int32_t N, int32_t H, int32_t W, int32_t C, void *stream);
/* x2 bilinear (align_corners=False) upsampling of a C-channel map times `scale`;
 * out is (N, 2H, 2W, out_cs); out2 (optional) receives a second copy (out2_cs) */
int dcvc_up2(const float *src, int32_t src_cs, float *out, int32_t out_cs, float *out2, int32_t out2_cs,
             int32_t N, int32_t H, int32_t W, int32_t C, float scale, void *stream);
/* 2x2 mean times `scale`; src is (N, H, W, .), H, W even.  avgpool_order = 0 sums in the order
 * of F.interpolate(bilinear, x0.5), 1 in the order of F.avg_pool2d(2, 2) (same value up to 1 ulp) */
int dcvc_down2(const float *src, int32_t src_cs, float *out, int32_t out_cs, int32_t N, int32_t H, int32_t W,
               int32_t C, float scale, int32_t avgpool_order, void *stream);
int dcvc_maxpool2(const float *src, int32_t src_cs, float *out, int32_t out_cs, int32_t N, int32_t H, int32_t W,
                  int32_t C, void *stream);
/* copy C channels between strided NHWC buffers */
int dcvc_copy_channels(const float *src, int32_t src_cs, float *out, int32_t out_cs, int64_t npix, int32_t C,
                       void *stream);
int dcvc_nchw_to_nhwc(const float *src, float *out, int32_t out_cs, int32_t N, int32_t C, int32_t H, int32_t W,
                      void *stream);
int dcvc_nhwc_to_nchw(const float *src, int32_t src_cs, float *out, int32_t N, int32_t C, int32_t H, int32_t W,
                      int32_t clamp01, void *stream);

/* ---- squeeze-excitation ---------------------------------------------------------------- */
/* mean(n, c) over H*W of a strided NHWC tensor (C a power of two in [4, 256]), deterministic
 * two-pass; scratch >= N*2048*C floats */
int dcvc_channel_mean(const float *src, int32_t src_cs, float *mean, float *scratch, int32_t N, int32_t HW,
                      int32_t C, void *stream);
/* gate = sigmoid(W2 relu(W1 mean)); W1: (Cr, C), W2: (C, Cr) as nn.Linear stores them */
int dcvc_se_gate(const float *mean, const float *w1, const float *w2, float *gate, int32_t N, int32_t C,
                 int32_t Cr, void *stream);

/* ---- quantisation / entropy-model elementwise ------------------------------------------ */
/* out = src / q  (mode 0) or src * q (mode 1);  q = max(q_basic[c], 0.5) * q_scale[n] */
int dcvc_scale_channels(const float *src, int32_t src_cs, float *out, int32_t out_cs, const float *q_basic,
                        const float *q_scale, int32_t mode, int32_t N, int32_t HW, int32_t C, void *stream);
/* z_hat = rint(z) (half to even); sym (optional): int32 in (n, c, y, x) order */
int dcvc_round_symbols(const float *z, int32_t z_cs, float *z_hat, int32_t zh_cs, int32_t *sym, int32_t N,
                       int32_t H, int32_t W, int32_t C, void *stream);
/* sym (n, c, y, x) int32 -> float NHWC */
int dcvc_symbols_to_nhwc(const int32_t *sym, float *out, int32_t out_cs, int32_t N, int32_t H, int32_t W,
                         int32_t C, void *stream);

typedef struct {
    const float *y;        /* (N,H,W,y_cs): latent already divided by curr_q (encoder only) */
    int32_t y_cs;
    const float *fusion;   /* (N,H,W,3C): [q_step | scales | means] from *_prior_fusion */
    int32_t fusion_cs;
    const float *spatial;  /* (N,H,W,2C): [scales_0 | means_0 | scales_1 | means_1]; step 2 only */
    int32_t spatial_cs;
    float *params;         /* (N,H,W,4C): [y_hat_0_0 | y_hat_1_1 | means | scales | q_step] */
    int32_t params_cs;
    float *y_hat;          /* (N,H,W,C) running y_hat in q_step units; step 2 finalises it */
    float *y_q;            /* (N,H,W,C) rounded residual (estimate path) or NULL */
    float *y_res;          /* (N,H,W,C) unrounded residual or NULL */
    float *scales_hat;     /* (N,H,W,C) masked scales or NULL */
    int32_t *sym;          /* (N, C/2, H, W) int32 symbols of this step or NULL */
    int32_t *idx;          /* (N, C/2, H, W) int32 CDF indexes of this step or NULL */
    float *out;            /* step 2: (N,H,W,out_cs) y_hat * q_step * curr_q */
    int32_t out_cs;
    const float *q_basic;  /* (C) */
    const float *q_scale;  /* (N) */
    int32_t N, H, W, C;
    int32_t step;          /* 0 or 1 */
    const float *idx_edges; /* device, 256 floats: edge[k-1] = smallest fp32 scale whose build_indexes value
                               (entropy_models.py:264-268, torch-CPU fp32) is >= k, for k = 1..255; edge[255] =
                               +inf.  index(s) = number of edges <= s.  Needed whenever idx is written. */
    const float *forced_q; /* encoder only, NULL in every product call: (N,H,W,C) rounded residuals to use INSTEAD of
                              round(y / q_step - mean) at the positions of this step -- "teacher forcing" with symbol
                              planes recorded from the reference, so that a parity test is not derailed by a value
                              that sits on a rounding tie (tests/test_gpu_backward.py, forced-symbol replay) */
} dcvc_dual_prior_args;

/* GaussianEncoder.build_indexes (entropy_models.py:264-268) on a flat array, bit-exact through idx_edges. */
int dcvc_scale_indexes(const float *scales, int32_t *idx, int64_t n, const float *idx_edges, void *stream);

int dcvc_dual_prior_enc(const dcvc_dual_prior_args *a, void *stream);
/* decoder step: idx only (sym == NULL) or apply decoded symbols (sym != NULL) */
int dcvc_dual_prior_dec_index(const dcvc_dual_prior_args *a, void *stream);
int dcvc_dual_prior_dec_apply(const dcvc_dual_prior_args *a, void *stream);

/* per-sample sums; out: (N) floats; scratch >= N*1024 floats.  kind: 0 laplace, 1 gaussian */
int dcvc_scale_bits(const float *y_q, const float *scales_hat, float *out, float *scratch, int32_t kind, int32_t N,
                    int64_t per_sample, void *stream);
/* factorised prior bits of z_hat (N,H,W,C; cs): params (11, C): h1,b1,a1,h2,b2,a2,h3,b3,a3,h4,b4 */
int dcvc_factorized_bits(const float *z_hat, int32_t z_cs, const float *params, float *out, float *scratch, int32_t N,
                         int32_t HW, int32_t C, void *stream);
/* sum over (y,x,c<C) of (a-b)^2 per sample */
int dcvc_sq_err(const float *a, int32_t a_cs, const float *b, int32_t b_cs, float *out, float *scratch, int32_t N,
                int32_t HW, int32_t C, void *stream);

/* ---- update(): entropy-model CDF tables built on the device (opt-in) -------------------------
 * GaussianEncoder.update (entropy_models.py:224-262) / BitEstimator.update (:119-174) incl. pmf_to_quantized_cdf
 * (ops.cpp:24-82).  Rows have dcvc_cdf_table_cols() int32 columns, zero-padded; sizes = symbols + 2, offsets =
 * -centre as CdfHelper stores them.  Device libm is not torch-CPU's: an entry whose probability sits on a
 * rounding boundary may differ by one count from the reference's table (valid, self-consistent, NOT for streams
 * a reference decoder must read -- those use the host builder, which is integer-identical). */
int32_t dcvc_cdf_table_cols(void);
/* scales: the 256 fp32 scale levels (device); kind 0 Laplace, 1 Gaussian */
int dcvc_build_scale_cdfs(const float *scales, int32_t n_scales, int32_t kind, int32_t *cdf, int32_t *sizes,
                          int32_t *offsets, void *stream);
/* params: (11, C) block as for dcvc_factorized_bits */
int dcvc_build_factorized_cdfs(const float *params, int32_t C, int32_t *cdf, int32_t *sizes, int32_t *offsets, void *stream);

const char *dcvc_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
