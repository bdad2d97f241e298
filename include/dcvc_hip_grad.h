/* dcvc_hip_grad.h -- C ABI of the backward (training) kernels for the DCVC-HEM P-frame path on
 * MI355X (gfx950).  Same conventions as dcvc_hip.h: raw device pointers, strided-NHWC fp32
 * activations, hipStream_t as void*, 0 / negative status, stream-ordered, caller-owned buffers.
 *
 * These are the kernels torch.autograd would run for the reference's training step
 * (/root/reference/core/model/dcvc_hem.py:104-587 calling DMC.forward_one_frame,
 * DCVC_HEM/src/models/video_model.py:470-596, then loss.backward()):
 *
 *   dcvc_conv_pack_weights_dev   device-side twin of dcvc_conv_pack_weights (weights change every
 *                                optimiser step); transposed=1 packs the flipped, channel-
 *                                transposed filter so that dcvc_conv2d computes the data gradient
 *   dcvc_pack_plan_create/_run/_destroy   the same packing jobs for every layer of a model in one launch
 *   dcvc_conv_bwd_prologue       backward of the fused epilogue of dcvc_conv2d: activation mask,
 *                                (gated) residual fan-out, inverse PixelShuffle, zero insertion
 *                                for stride-2 layers                layers.py:18-127
 *   dcvc_conv_wgrad              dL/dW of nn.Conv2d (MFMA, pixels as the reduction dimension)
 *   dcvc_channel_dot             per-channel sums of a (.* b): bias gradients, SE gate gradients,
 *                                q-scale gradients
 *   dcvc_mask_accumulate         dst += src * leaky'(x): backward of an activation-on-load
 *   dcvc_warp_bwd                F.grid_sample(bilinear, border, align_corners) backward
 *                                                                  video_net.py:32-55
 *   dcvc_up2_bwd / dcvc_down2_bwd / dcvc_maxpool2_bwd               video_net.py:58-71,132-133,185
 *   dcvc_se_bwd, dcvc_add_channel_vec   SELayer backward            video_net.py:149-162
 *   dcvc_scale_channels_bwd, dcvc_q_finish   y / curr_q, y_hat * curr_q and LowerBound(q_basic)
 *                                                                  video_model.py:255-261, video_net.py:14-28
 *   dcvc_dual_prior_bwd          forward_dual_prior backward (straight-through round)
 *                                                                  common_model.py:38-49,82-177
 *   dcvc_scale_bits_bwd / dcvc_factorized_bits_bwd / dcvc_sq_err_bwd
 *                                get_y_laplace_bits, get_z_bits + Bitparm, probs_to_bits with
 *                                LowerBound, MSE                    common_model.py:51-73
 *   dcvc_add_planes              x + noise (add_noise)              common_model.py:46-49
 */
#ifndef DCVC_HIP_GRAD_H
#define DCVC_HIP_GRAD_H

#include <stdint.h>

#include "dcvc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* w: DEVICE (Cout, Cin_total, ks, ks); b: DEVICE (Cout) or NULL.
 * transposed = 0: pack input channels [cin_offset, cin_offset + sum(seg_C)) exactly like
 *   dcvc_conv_pack_weights (seg_C on the host).
 * transposed = 1: nseg must be 1; packs the filter of the data-gradient convolution whose input
 *   channels are the Cout channels of the forward layer and whose output channels are forward
 *   input channels [cin_offset, cin_offset + seg_C[0]):  w'[ci][co][ky][kx] = w[co][ci][K-1-ky][K-1-kx];
 *   bpack is zero-filled.  wpack/bpack sizes: dcvc_conv_pack_size(seg_C[0], ks, 1, {Cout}, ..). */
int dcvc_conv_pack_weights_dev(const float *w, const float *b, int32_t Cout, int32_t Cin_total, int32_t ks,
                               int32_t nseg, const int32_t *seg_C, int32_t cin_offset, int32_t pixel_shuffle,
                               int32_t precision, int32_t transposed, float *wpack, float *bpack, void *stream);

/* The same jobs for every layer of a model in ONE launch (a training step re-packs ~370 filters after each optimiser
 * step).  A job is the argument list of dcvc_conv_pack_weights_dev; the plan keeps the jobs on the device.
 * create: synchronous (allocates, copies the table); run: one stream-ordered launch; the pointers in the jobs must stay
 * valid for the plan's life.  Results are identical to the per-layer calls. */
typedef struct {
    const float *w, *b;
    int32_t Cout, Cin_total, ks, nseg;
    int32_t seg_C[DCVC_MAX_SEG];
    int32_t cin_offset, pixel_shuffle, precision, transposed;
    float *wpack, *bpack;
} dcvc_pack_job;
int dcvc_pack_plan_create(const dcvc_pack_job *jobs, int32_t n, void **plan);
int dcvc_pack_plan_run(void *plan, void *stream);
void dcvc_pack_plan_destroy(void *plan);

typedef struct {
    const float *dout;    /* gradient of the layer's output view (N, Ho*m, Wo*m, .), m = 2 if pixel_shuffle */
    int32_t dout_cs;
    const float *out;     /* the forward output (needed for the activation mask); may be NULL if act == 0 */
    int32_t out_cs;
    const float *res;     /* forward residual / gate / second residual as given to dcvc_conv2d */
    int32_t res_cs;
    const float *gate;
    const float *res2;
    int32_t res2_cs;
    float *dres;          /* += dout * gate   (NULL: not needed) */
    int32_t dres_cs;
    float *dres2;         /* += dout          (NULL: not needed) */
    int32_t dres2_cs;
    float *dpre;          /* gradient before bias/activation in nn.Conv2d channel order, written at
                             pixel (oy*zs, ox*zs) of an (N, Hd, Wd, dpre_cs) buffer (pre-zeroed if zs > 1) */
    int32_t dpre_cs, zs, Hd, Wd;
    int32_t N, Ho, Wo, Cout; /* conv output geometry before pixel shuffle */
    int32_t pixel_shuffle;
    int32_t act;          /* 1: LeakyReLU(slope) was applied after bias */
    float slope;
} dcvc_conv_bwd_args;

int dcvc_conv_bwd_prologue(const dcvc_conv_bwd_args *a, void *stream);

typedef struct {
    const float *x;       /* one input segment (N, Hin, Win, x_cs), C channels */
    int32_t x_cs, C;
    int32_t in_act;       /* activation applied on load in the forward pass */
    float in_slope;
    const float *dpre;    /* (N, Hd, Wd, dpre_cs): gradient of output pixel (oy, ox) at (oy*zs, ox*zs) */
    int32_t dpre_cs, zs, Hd, Wd;
    int32_t N, Hin, Win, Ho, Wo, Cout, ks, stride;
    float *dw;            /* (Cout, Cin_total, ks, ks), accumulated (+=) for ci in [cin_offset, cin_offset+C) */
    int32_t Cin_total, cin_offset;
    float *scratch;       /* partial sums; at least dcvc_conv_wgrad_scratch_min floats */
    int64_t scratch_floats;
    int32_t overwrite;    /* 1: write (=) the dw slice instead of accumulating: no zero fill needed */
    float *db;            /* optional (Cout): bias gradient = sum of dpre over samples and pixels, written (=)
                             from the same pass over dpre */
    int32_t precision;    /* DCVC_PREC_FP32: fp32 MFMA (exact fmaf chains).  DCVC_PREC_FP16X3 (the engine's fast mode),
                             stride-1 layers: operands split into bf16 hi + lo, three bf16 MFMAs per product, fp32
                             accumulation -- relative error ~2^-16 of sum |dY||X|; stride-2 layers stay fp32 */
} dcvc_conv_wgrad_args;

int64_t dcvc_conv_wgrad_scratch_min(int32_t Cout, int32_t C, int32_t ks);
int dcvc_conv_wgrad(const dcvc_conv_wgrad_args *a, void *stream);

/* out (+)= sum over pixels of a * b (b == NULL: of a).  over_batch = 0: out is (N, C);
 * over_batch = 1: out is (C), summed over samples too.  accumulate: += instead of =.
 * scratch >= N * 256 * round4(C) floats.  Deterministic (fixed-order partial sums). */
int dcvc_channel_dot(const float *a, int32_t a_cs, const float *b, int32_t b_cs, float *out, float *scratch,
                     int32_t N, int32_t HW, int32_t C, int32_t over_batch, int32_t accumulate, void *stream);

/* dst += src * (x > 0 ? 1 : slope)   (x == NULL: dst += src) */
int dcvc_mask_accumulate(const float *src, int32_t src_cs, const float *x, int32_t x_cs, float slope, float *dst,
                         int32_t dst_cs, int64_t npix, int32_t C, void *stream);
/* out = a + b (dense or strided planes) */
int dcvc_add_planes(const float *a, int32_t a_cs, const float *b, int32_t b_cs, float *out, int32_t out_cs,
                    int64_t npix, int32_t C, void *stream);

/* dsrc (+=, may be NULL) and dflow (+=, 2 channels, may be NULL) of dcvc_warp.  The source scatter is summed in
 * 64-bit fixed point so that the result is independent of the order the atomics land in; the step is 2^-40 of the
 * largest |dout| of THIS call (found by a first pass), so gradients of any magnitude keep ~fp32 relative accuracy, and
 * a non-finite dout makes every dsrc element NaN instead of a clamped finite value:
 * fix_scratch = N*H*W*C + 1 8-byte words, ALL ZERO on entry, left all zero on return (required when dsrc != NULL). */
int dcvc_warp_bwd(const float *src, int32_t src_cs, const float *flow, int32_t flow_cs, const float *dout,
                  int32_t dout_cs, float *dsrc, int32_t dsrc_cs, float *dflow, int32_t dflow_cs, int32_t N, int32_t H,
                  int32_t W, int32_t C, void *fix_scratch, void *stream);
/* dsrc += adjoint of dcvc_up2 (dout is (N, 2H, 2W, .)); gathered per source element, no atomics */
int dcvc_up2_bwd(const float *dout, int32_t dout_cs, float *dsrc, int32_t dsrc_cs, int32_t N, int32_t H, int32_t W,
                 int32_t C, float scale, void *stream);
/* dsrc (N, H, W, .) += scale/4 * dout(y/2, x/2) */
int dcvc_down2_bwd(const float *dout, int32_t dout_cs, float *dsrc, int32_t dsrc_cs, int32_t N, int32_t H, int32_t W,
                   int32_t C, float scale, void *stream);
/* dsrc[argmax of each 2x2 window (first maximum in scan order)] += dout */
int dcvc_maxpool2_bwd(const float *src, int32_t src_cs, const float *dout, int32_t dout_cs, float *dsrc,
                      int32_t dsrc_cs, int32_t N, int32_t H, int32_t W, int32_t C, void *stream);

/* SELayer FC part: given dgate (N, C) returns dmean (N, C) and accumulates dw1 (Cr, C), dw2 (C, Cr) */
int dcvc_se_bwd(const float *mean, const float *w1, const float *w2, const float *gate, const float *dgate,
                float *dmean, float *dw1, float *dw2, int32_t N, int32_t C, int32_t Cr, void *stream);
/* dst(n, p, c) += vec(n, c) * scale */
int dcvc_add_channel_vec(float *dst, int32_t dst_cs, const float *vec, float scale, int32_t N, int32_t HW, int32_t C,
                         void *stream);

/* backward of dcvc_scale_channels: dsrc += dout / q (mode 0) or dout * q (mode 1), q as there */
int dcvc_scale_channels_bwd(const float *dout, int32_t dout_cs, float *dsrc, int32_t dsrc_cs, const float *q_basic,
                            const float *q_scale, int32_t mode, int32_t N, int32_t HW, int32_t C, void *stream);
/* Gradient of curr_q(n, c) = max(q_basic[c], 0.5) * q_scale[n] collected over all its uses:
 *   dq(n, c) = dq_mul(n, c) - s_div(n, c) / curr_q(n, c)
 * where dq_mul holds sums of dout * (d out / d curr_q) from multiplications by curr_q and s_div
 * sums of dout * out from divisions by it (either may be NULL).  Then LowerBound(q_basic, 0.5)
 * backward (video_net.py:14-28) and the product rule:
 *   dq_basic[c] += pass(c) * sum_n dq(n,c) q_scale[n],  dq_scale[n] += sum_c dq(n,c) max(q_basic[c], 0.5).
 * Either output may be NULL. */
int dcvc_q_finish(const float *dq_mul, const float *s_div, const float *q_basic, const float *q_scale,
                  float *dq_basic, float *dq_scale, int32_t N, int32_t C, void *stream);

typedef struct {
    const float *y;         /* forward inputs, as dcvc_dual_prior_args */
    int32_t y_cs;
    const float *fusion;
    int32_t fusion_cs;
    const float *y_hat;     /* (N,H,W,C) dense: final y_hat in q_step units (forward output) */
    const float *dout;      /* gradient of args.out (y_hat * q_step * curr_q) */
    int32_t dout_cs;
    const float *dy_res;    /* (N,H,W,C) dense gradients of the y_res / scales_hat planes */
    const float *dscales_hat;
    const float *dparams;   /* step 0 only: gradient of the (N,H,W,4C) params buffer */
    int32_t dparams_cs;
    float *dspatial;        /* step 1: (N,H,W,2C) written (=) */
    int32_t dspatial_cs;
    float *dy;              /* step 0: += */
    int32_t dy_cs;
    float *dfusion;         /* step 0: += (N,H,W,3C) */
    int32_t dfusion_cs;
    float *dq_plane;        /* step 0: (N,H,W,C) dense, = dout * y_hat * q_step (to be reduced into dq) */
    const float *q_basic;
    const float *q_scale;
    int32_t N, H, W, C;
    int32_t step;           /* run step 1 first, then (after the spatial prior's backward) step 0 */
} dcvc_dual_prior_bwd_args;

int dcvc_dual_prior_bwd(const dcvc_dual_prior_bwd_args *a, void *stream);

/* g: (N) upstream gradient of the per-sample sums of dcvc_scale_bits (kind 0: Laplace, get_y_laplace_bits,
 * common_model.py:64-69; kind 1: Gaussian, get_y_gaussian_bits, :57-62 -- IntraNoAR's training mode, round 4);
 * dy, dscales: dense planes, written (=) */
int dcvc_scale_bits_bwd(const float *y, const float *scales_hat, const float *g, float *dy, float *dscales,
                        int32_t kind, int32_t N, int64_t per_sample, void *stream);
/* dz (strided, +=) and dparams (11, C) += of dcvc_factorized_bits evaluated at z */
int dcvc_factorized_bits_bwd(const float *z, int32_t z_cs, const float *params, const float *g, float *dz,
                             int32_t dz_cs, float *dparams, int32_t N, int32_t HW, int32_t C, void *stream);
/* da += 2 (a - b) g[n] */
int dcvc_sq_err_bwd(const float *a, int32_t a_cs, const float *b, int32_t b_cs, const float *g, float *da,
                    int32_t da_cs, int32_t N, int32_t HW, int32_t C, void *stream);

#ifdef __cplusplus
}
#endif
#endif
