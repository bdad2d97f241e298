// entropy_kernels.hip -- the elementwise / reduction side of the hyperprior + dual-prior entropy
// model, fused into a handful of HBM-bound kernels (the reference issues ~40 tiny ATen
// kernels per dual-prior call and rebuilds the checkerboard mask every time).
//
//   dual_prior_*      CompressionModel.forward_dual_prior / process_with_mask / get_mask /
//                     decompress_dual_prior   /root/reference/DCVC_HEM/src/models/common_model.py:82-217
//   scale index       GaussianEncoder.build_indexes  src/entropy_models/entropy_models.py:264-268
//   *_bits            get_y_laplace_bits / get_y_gaussian_bits / get_z_bits / probs_to_bits
//                     common_model.py:51-73, BitEstimator/Bitparm entropy_models.py:54-117
//   channel_mean/se   SELayer  src/models/video_net.py:149-162
//
// Reductions are two-pass with fixed block counts and in-order final sums, so results are
// bit-identical run to run (no atomics): the decoder must re-derive the encoder's numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dcvc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

#define RET_LAUNCH() return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH
inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

constexpr int RB = 1024;  // partial-sum blocks per sample for scalar reductions
constexpr int MB = 2048;  // partial-sum blocks per sample for channel means

__device__ __forceinline__ float block_sum(float v, float *sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) sm[t] += sm[t + s];
        __syncthreads();
    }
    return sm[0];
}

__global__ void finish_sum(const float *__restrict__ scratch, float *__restrict__ out, int nb) {
    __shared__ float sm[256];
    const int n = blockIdx.x;
    float v = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) v += scratch[(size_t)n * nb + i];
    const float s = block_sum(v, sm);
    if (threadIdx.x == 0) out[n] = s;
}

// ---- SE ------------------------------------------------------------------------------------
// Per-channel partial sums over a slice of pixels: 16 B per lane (4 channels), MB blocks per
// sample so the read runs at HBM rate; partials are reduced in a fixed order (no atomics).
__global__ void channel_partial(const float *__restrict__ src, int cs, float *__restrict__ scratch, int HW, int C) {
    __shared__ f32x4 sm[256];
    const int n = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    const int C4 = C >> 2;           // lanes per pixel
    const int G = 256 / C4;          // pixels per block iteration
    const int c4 = t % C4, g = t / C4;
    const int per = (HW + MB - 1) / MB;
    const int p0 = b * per, p1 = min(HW, p0 + per);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (g < G)
        for (int p = p0 + g; p < p1; p += G) v += *(const f32x4 *)&src[((size_t)n * HW + p) * cs + c4 * 4];
    sm[t] = v;
    __syncthreads();
    if (t < C4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < G; ++k) s += sm[k * C4 + t];
        *(f32x4 *)&scratch[((size_t)n * MB + b) * C + t * 4] = s;
    }
}

// grid (N, ceil(C/16)): a block finishes 16 channels with 16 interleaved partial sums each
// (fixed order, so the result is run-to-run identical).
__global__ void channel_finish(const float *__restrict__ scratch, float *__restrict__ mean, int HW, int C) {
    __shared__ float sm[256];
    const int n = blockIdx.x, t = threadIdx.x;
    const int c = blockIdx.y * 16 + (t & 15), part = t >> 4;
    float s = 0.f;
    if (c < C)
        for (int b = part; b < MB; b += 16) s += scratch[((size_t)n * MB + b) * C + c];
    sm[t] = s;
    __syncthreads();
    if (t < 16 && c < C) {
        float r = 0.f;
        for (int k = 0; k < 16; ++k) r += sm[k * 16 + t];
        mean[(size_t)n * C + c] = r / (float)HW;
    }
}

// the same finish over the partial rows a convolution's epilogue wrote (dcvc_conv_args.chan_partial): a 1080p
// layer leaves 8160 rows, so 64 row-lanes per channel with four independent accumulators each keep enough loads
// in flight (16 row-lanes took 86 us); fixed assignment and fixed tree -> run-to-run identical
__global__ __launch_bounds__(1024) void channel_finish_rows(const float *__restrict__ part, int parts, int stride,
                                                            float *__restrict__ mean, int HW, int C) {
    __shared__ float sm[1024];
    const int n = blockIdx.x, t = threadIdx.x;
    const int c = blockIdx.y * 16 + (t & 15), rl = t >> 4;  // 64 row-lanes
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        const float *p = part + (size_t)n * parts * stride + c;
        int b = rl;
        for (; b + 192 < parts; b += 256) {
            s0 += p[(size_t)b * stride];
            s1 += p[(size_t)(b + 64) * stride];
            s2 += p[(size_t)(b + 128) * stride];
            s3 += p[(size_t)(b + 192) * stride];
        }
        for (; b < parts; b += 64) s0 += p[(size_t)b * stride];
    }
    sm[t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (t < 16 && c < C) {
        float r = 0.f;
        for (int k = 0; k < 64; ++k) r += sm[k * 16 + t];
        mean[(size_t)n * C + c] = r / (float)HW;
    }
}

__global__ void se_gate_kernel(const float *__restrict__ mean, const float *__restrict__ w1,
                               const float *__restrict__ w2, float *__restrict__ gate, int C, int Cr) {
    __shared__ float hid[64];
    const int n = blockIdx.x, t = threadIdx.x;
    if (t < Cr) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += w1[t * C + c] * mean[(size_t)n * C + c];
        hid[t] = fmaxf(s, 0.f);
    }
    __syncthreads();
    if (t < C) {
        float s = 0.f;
        for (int j = 0; j < Cr; ++j) s += w2[t * Cr + j] * hid[j];
        gate[(size_t)n * C + t] = 1.f / (1.f + expf(-s));
    }
}

// ---- quantisation --------------------------------------------------------------------------
__global__ void scale_channels_kernel(const float *__restrict__ src, int src_cs, float *__restrict__ out, int out_cs,
                                      const float *__restrict__ q_basic, const float *__restrict__ q_scale, int mode,
                                      int64_t HW, int C, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int n = (int)(pix / HW);
    const float q = fmaxf(q_basic[c], 0.5f) * q_scale[n];
    const float v = src[pix * src_cs + c];
    out[pix * out_cs + c] = mode ? v * q : v / q;
}

__global__ void round_symbols_kernel(const float *__restrict__ z, int z_cs, float *__restrict__ zh, int zh_cs,
                                     int32_t *__restrict__ sym, int H, int W, int C, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    const float r = rintf(z[pix * z_cs + c]);
    if (zh) zh[pix * zh_cs + c] = r;
    if (sym) sym[((n * C + c) * H + y) * (int64_t)W + x] = (int32_t)r;
}

__global__ void symbols_to_nhwc_kernel(const int32_t *__restrict__ sym, float *__restrict__ out, int out_cs, int H,
                                       int W, int C, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    out[pix * out_cs + c] = (float)sym[((n * C + c) * H + y) * (int64_t)W + x];
}

// GaussianEncoder.build_indexes (entropy_models.py:264-268) without a device logarithm: the index is
// a non-decreasing step function of the scale, so it equals the number of bin edges <= s, where
// edge[k-1] is the smallest fp32 scale whose reference (torch-CPU fp32) index is >= k.  The 255
// edges are found on the host by bisection over the float's bit pattern with the reference formula
// itself (vcm_ts_amd/entropy.py scale_index_edges), so encoder, decoder and reference agree bit
// for bit.  `edges` has 256 entries, the last one +inf.  Scales below 1e-5 (incl. negatives) sit
// under every edge -> 0, like the reference's clamp; a NaN compares false everywhere -> 0.
__device__ __forceinline__ int32_t scale_index(float s, const float *edges) {
    int lo = 0;  // invariant: edges[0..lo) <= s
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1)
        if (edges[lo + step - 1] <= s) lo += step;
    return lo;
}

__global__ void scale_indexes_kernel(const float *__restrict__ scales, int32_t *__restrict__ idx, int64_t n,
                                     const float *__restrict__ edges_g) {
    __shared__ float edges[256];
    edges[threadIdx.x] = edges_g[threadIdx.x];
    __syncthreads();
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n) idx[gid] = scale_index(scales[gid], edges);
}

// mode 0: encoder (needs y); mode 1: decoder index pass; mode 2: decoder apply pass
template <int MODE>
__global__ void dual_prior_kernel(const dcvc_dual_prior_args a, int64_t total) {
    __shared__ float edges[256];
    if (MODE != 2) {  // the apply pass of the decoder writes no indexes
        edges[threadIdx.x] = a.idx_edges ? a.idx_edges[threadIdx.x] : 0.f;
        __syncthreads();
    }
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int C = a.C, Ch = C >> 1;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int x = (int)(pix % a.W), y = (int)((pix / a.W) % a.H);
    const int64_t n = pix / ((int64_t)a.W * a.H);
    const bool half = c >= Ch;
    const int k = half ? c - Ch : c;
    const bool m0 = ((x + y) & 1) == 0;
    const bool active = a.step == 0 ? (half ? !m0 : m0) : (half ? m0 : !m0);
    const float *fu = a.fusion + pix * a.fusion_cs;
    const float qs = fmaxf(fu[c], 0.5f);
    float sc, mu;
    if (a.step == 0) {
        sc = fu[C + c];
        mu = fu[2 * C + c];
    } else {
        const float *sp = a.spatial + pix * a.spatial_cs;
        sc = sp[half ? C + k : k];
        mu = sp[half ? C + Ch + k : Ch + k];
    }
    const int64_t e = pix * C + c;  // dense NHWC index of the C-channel planes
    const int64_t s_i = ((n * Ch + k) * a.H + y) * (int64_t)a.W + x;
    if (MODE == 1) {
        if (a.step == 0) {
            float *pr = a.params + pix * a.params_cs;
            pr[C + c] = fu[2 * C + c];
            pr[2 * C + c] = fu[C + c];
            pr[3 * C + c] = qs;
        }
        if (active) a.idx[s_i] = scale_index(sc, edges);
        return;
    }
    float hat = 0.f;
    if (active) {
        float q;
        if (MODE == 0) {
            const float yq = a.y[pix * a.y_cs + c] / qs;
            const float res = yq - mu;
            q = a.forced_q ? a.forced_q[e] : rintf(res);
            if (a.y_res) a.y_res[e] = res;
            if (a.y_q) a.y_q[e] = q;
            if (a.scales_hat) a.scales_hat[e] = sc;
            if (a.sym) a.sym[s_i] = (int32_t)q;
            if (a.idx) a.idx[s_i] = scale_index(sc, edges);
        } else {
            q = (float)a.sym[s_i];
        }
        hat = q + mu;
        a.y_hat[e] = hat;
    } else if (a.step == 1) {
        hat = a.y_hat[e];
    }
    if (a.step == 0) {
        float *pr = a.params + pix * a.params_cs;
        pr[c] = hat;
        if (MODE == 0) {
            pr[C + c] = mu;
            pr[2 * C + c] = sc;
            pr[3 * C + c] = qs;
        }
    } else {
        const float cq = fmaxf(a.q_basic[c], 0.5f) * a.q_scale[n];
        a.out[pix * a.out_cs + c] = (hat * qs) * cq;
    }
}

// ---- rate / distortion sums ------------------------------------------------------------------
__device__ __forceinline__ float bits_of(float p) { return fmaxf((-1.0f * logf(p + 1e-5f)) / 0.6931471805599453f, 0.f); }

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

__device__ __forceinline__ float laplace_cdf(float t, float b) { return 0.5f - 0.5f * sgn(t) * expm1f(-fabsf(t) / b); }

__device__ __forceinline__ float normal_cdf(float t, float sigma) {
    return 0.5f * (1.f + erff(t * (1.f / sigma) / 1.4142135623730951f));
}

__global__ void scale_bits_kernel(const float *__restrict__ yq, const float *__restrict__ sh, float *__restrict__ scratch,
                                  int kind, int64_t per) {
    __shared__ float sm[256];
    const int n = blockIdx.y;
    const float *a = yq + (size_t)n * per, *b = sh + (size_t)n * per;
    float v = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)RB * 256) {
        const float y = a[i];
        float p;
        if (kind == 0) {
            const float s = fminf(fmaxf(b[i], 1e-5f), 1e10f);
            p = laplace_cdf(y + 0.5f, s) - laplace_cdf(y - 0.5f, s);
        } else {
            const float s = fminf(fmaxf(b[i], 0.11f), 1e10f);
            p = normal_cdf(y + 0.5f, s) - normal_cdf(y - 0.5f, s);
        }
        v += bits_of(p);
    }
    const float s = block_sum(v, sm);
    if (threadIdx.x == 0) scratch[(size_t)n * RB + blockIdx.x] = s;
}

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__device__ __forceinline__ float fact_cdf(float x, const float *__restrict__ P, int C, int c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x = x * softplusf(P[(3 * i) * C + c]) + P[(3 * i + 1) * C + c];
        x = x + tanhf(x) * tanhf(P[(3 * i + 2) * C + c]);
    }
    x = x * softplusf(P[9 * C + c]) + P[10 * C + c];
    return 1.f / (1.f + expf(-x));
}

__global__ void factorized_bits_kernel(const float *__restrict__ z, int z_cs, const float *__restrict__ P,
                                       float *__restrict__ scratch, int64_t HW, int C) {
    __shared__ float sm[256];
    const int n = blockIdx.y;
    const int64_t per = HW * C;
    float v = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)RB * 256) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const float zz = z[((size_t)n * HW + pix) * z_cs + c];
        v += bits_of(fact_cdf(zz + 0.5f, P, C, c) - fact_cdf(zz - 0.5f, P, C, c));
    }
    const float s = block_sum(v, sm);
    if (threadIdx.x == 0) scratch[(size_t)n * RB + blockIdx.x] = s;
}

__global__ void sq_err_kernel(const float *__restrict__ a, int a_cs, const float *__restrict__ b, int b_cs,
                              float *__restrict__ scratch, int64_t HW, int C) {
    __shared__ float sm[256];
    const int n = blockIdx.y;
    const int64_t per = HW * C;
    float v = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)RB * 256) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const float d = a[((size_t)n * HW + pix) * a_cs + c] - b[((size_t)n * HW + pix) * b_cs + c];
        v += d * d;
    }
    const float s = block_sum(v, sm);
    if (threadIdx.x == 0) scratch[(size_t)n * RB + blockIdx.x] = s;
}

// ---- update(): CDF tables on the device (opt-in) -------------------------------------------
// GaussianEncoder.update (entropy_models.py:224-262) and BitEstimator.update (:119-174) with the quantiser of
// EntropyCoder.pmf_to_quantized_cdf (ops.cpp:24-82).  The reference evaluates the Laplace / Normal /
// factorised CDFs with torch-CPU fp32 kernels (SLEEF); the device's expm1f / erff / expf / tanhf are not the
// same functions to the last ulp, so a probability next to a rounding boundary of round(p * 2^16) can land in
// the neighbouring integer: these tables are valid and self-consistent (encoder and decoder of THIS library
// agree), but interoperable streams use the host-built tables, which are integer-identical to the reference.
constexpr int kTabCols = 104;  // 2 * 50 + 1 symbols + tail bin + 1, rounded up

__device__ void quantize_row(const float *pmf, int n, uint32_t *cdf) {  // ops.cpp:24-82, serial
    const float scale = 65536.f;
    uint32_t total = 0;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) {
        cdf[i + 1] = (uint32_t)(roundf(pmf[i] * scale) + 0.5f);
        total += cdf[i + 1];
    }
    uint32_t run = 0;
    for (int i = 0; i <= n; ++i) {
        run += (uint32_t)((((uint64_t)1 << 16) * cdf[i]) / (total ? total : 1));
        cdf[i] = run;
    }
    cdf[n] = 1u << 16;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best = ~0u;
        int donor = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t w = cdf[j + 1] - cdf[j];
            if (w > 1 && w < best) best = w, donor = j;
        }
        if (donor < 0) return;
        if (donor < i)
            for (int j = donor + 1; j <= i; ++j) cdf[j]--;
        else
            for (int j = i + 1; j <= donor; ++j) cdf[j]++;
    }
}

// one block per scale level
__global__ void scale_cdfs_kernel(const float *__restrict__ scales, int kind, int32_t *__restrict__ cdf,
                                  int32_t *__restrict__ sizes, int32_t *__restrict__ offsets) {
    __shared__ float pmf[kTabCols];
    __shared__ uint32_t q[kTabCols + 1];
    __shared__ int center;
    const int row = blockIdx.x, t = threadIdx.x;
    const float s = scales[row];
    auto F = [&](float x) { return kind == 0 ? laplace_cdf(x, s) : normal_cdf(x, s); };
    if (t == 0) center = 50;
    __syncthreads();
    if (t >= 2 && t <= 50 && F((float)t) > 0.9999f) atomicMin(&center, t);  // smallest i whose tail is below 1e-4
    __syncthreads();
    const int c = center, len = 2 * c + 1;
    if (t < len) {
        const float x = (float)(t - c);
        pmf[t] = F(x + 0.5f) - F(x - 0.5f);
    }
    if (t == len) pmf[len] = 2.f * F((float)(-c) - 0.5f);  // tail mass = 2 * lower[0]
    for (int i = t; i <= kTabCols; i += blockDim.x) q[i] = 0;
    __syncthreads();
    if (t == 0) {
        quantize_row(pmf, len + 1, q);
        sizes[row] = len + 2;
        offsets[row] = -c;
    }
    __syncthreads();
    for (int i = t; i < kTabCols; i += blockDim.x) cdf[row * kTabCols + i] = i <= len + 1 ? (int32_t)q[i] : 0;
}

// one block for all channels of a factorised prior (C <= 256)
__global__ void factorized_cdfs_kernel(const float *__restrict__ P, int C, int32_t *__restrict__ cdf,
                                       int32_t *__restrict__ sizes, int32_t *__restrict__ offsets) {
    extern __shared__ unsigned char smem[];
    int *lo = (int *)smem, *hi = lo + C;
    int *max_len = hi + C;
    float *pmf = (float *)(max_len + 4);
    const int t = threadIdx.x;
    for (int c = t; c < C; c += blockDim.x) lo[c] = 50, hi[c] = 50;
    if (t == 0) *max_len = 0;
    __syncthreads();
    for (int k = t; k < C * 49; k += blockDim.x) {
        const int c = k / 49, i = 2 + k % 49;
        if (fact_cdf(-(float)i, P, C, c) < 0.0001f) atomicMin(&lo[c], i);
        if (fact_cdf((float)i, P, C, c) > 0.9999f) atomicMin(&hi[c], i);
    }
    __syncthreads();
    for (int c = t; c < C; c += blockDim.x) atomicMax(max_len, hi[c] + lo[c] + 1);
    __syncthreads();
    const int ml = *max_len;
    for (int k = t; k < C * kTabCols; k += blockDim.x) {
        const int c = k / kTabCols, j = k % kTabCols, len = hi[c] + lo[c] + 1;
        float v = 0.f;
        if (j < len) {
            const float x = (float)(j - lo[c]);
            v = fact_cdf(x + 0.5f, P, C, c) - fact_cdf(x - 0.5f, P, C, c);
        } else if (j == len) {  // tail: lower[0] + (1 - upper[max_len - 1]) over the PADDED sample range (:160-164)
            v = fact_cdf(-(float)lo[c] - 0.5f, P, C, c) + (1.0f - fact_cdf((float)(ml - 1 - lo[c]) + 0.5f, P, C, c));
        }
        pmf[k] = v;
    }
    __syncthreads();
    for (int c = t; c < C; c += blockDim.x) {
        const int len = hi[c] + lo[c] + 1;
        uint32_t q[kTabCols + 1];
        for (int i = 0; i <= kTabCols; ++i) q[i] = 0;
        quantize_row(pmf + c * kTabCols, len + 1, q);
        for (int i = 0; i < kTabCols; ++i) cdf[c * kTabCols + i] = i <= len + 1 ? (int32_t)q[i] : 0;
        sizes[c] = len + 2;
        offsets[c] = -lo[c];
    }
}

}  // namespace

extern "C" int dcvc_channel_mean(const float *src, int32_t src_cs, float *mean, float *scratch, int32_t N, int32_t HW,
                                 int32_t C, void *stream) {
    if (!src || !mean || !scratch || N <= 0 || HW <= 0 || C < 4 || C > 256 || (1024 % C) || (src_cs & 3) ||
        ((uintptr_t)src & 15))
        return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(channel_partial, dim3(MB, N), dim3(256), 0, st, src, src_cs, scratch, HW, C);
    hipLaunchKernelGGL(channel_finish, dim3(N, (C + 15) / 16), dim3(256), 0, st, scratch, mean, HW, C);
    RET_LAUNCH();
}

extern "C" int dcvc_channel_mean_finish(const float *chan_partial, int32_t parts, int32_t row_stride, float *mean, int32_t N,
                                        int32_t C, int32_t HW, void *stream) {
    if (!chan_partial || !mean || parts <= 0 || row_stride < C || N <= 0 || C <= 0 || HW <= 0) return DCVC_E_ARG;
    hipLaunchKernelGGL(channel_finish_rows, dim3(N, (C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, chan_partial, parts,
                       row_stride, mean, HW, C);
    RET_LAUNCH();
}

extern "C" int dcvc_se_gate(const float *mean, const float *w1, const float *w2, float *gate, int32_t N, int32_t C,
                            int32_t Cr, void *stream) {
    if (!mean || !w1 || !w2 || !gate || C > 256 || Cr > 64 || Cr <= 0) return DCVC_E_ARG;
    hipLaunchKernelGGL(se_gate_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, mean, w1, w2, gate, C, Cr);
    RET_LAUNCH();
}

extern "C" int dcvc_scale_channels(const float *src, int32_t src_cs, float *out, int32_t out_cs, const float *q_basic,
                                   const float *q_scale, int32_t mode, int32_t N, int32_t HW, int32_t C, void *stream) {
    if (!src || !out || !q_basic || !q_scale) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * HW * C;
    hipLaunchKernelGGL(scale_channels_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, src, src_cs, out,
                       out_cs, q_basic, q_scale, mode, (int64_t)HW, C, total);
    RET_LAUNCH();
}

extern "C" int dcvc_round_symbols(const float *z, int32_t z_cs, float *z_hat, int32_t zh_cs, int32_t *sym, int32_t N,
                                  int32_t H, int32_t W, int32_t C, void *stream) {
    if (!z || (!z_hat && !sym)) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(round_symbols_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, z, z_cs, z_hat,
                       zh_cs, sym, H, W, C, total);
    RET_LAUNCH();
}

extern "C" int dcvc_symbols_to_nhwc(const int32_t *sym, float *out, int32_t out_cs, int32_t N, int32_t H, int32_t W,
                                    int32_t C, void *stream) {
    if (!sym || !out) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(symbols_to_nhwc_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, sym, out,
                       out_cs, H, W, C, total);
    RET_LAUNCH();
}

static int dual_prior_check(const dcvc_dual_prior_args *a) {
    if (!a || !a->fusion || !a->params || !a->y_hat || (a->C & 1) || a->N <= 0) return DCVC_E_ARG;
    if (a->step != 0 && a->step != 1) return DCVC_E_ARG;
    if (a->step == 1 && (!a->spatial || !a->out || !a->q_basic || !a->q_scale)) return DCVC_E_ARG;
    return DCVC_OK;
}

extern "C" int dcvc_scale_indexes(const float *scales, int32_t *idx, int64_t n, const float *idx_edges, void *stream) {
    if (!scales || !idx || !idx_edges || n < 0) return DCVC_E_ARG;
    if (n == 0) return DCVC_OK;
    hipLaunchKernelGGL(scale_indexes_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, scales, idx, n,
                       idx_edges);
    RET_LAUNCH();
}

extern "C" int dcvc_dual_prior_enc(const dcvc_dual_prior_args *a, void *stream) {
    if (dual_prior_check(a) || !a->y || (a->idx && !a->idx_edges)) return DCVC_E_ARG;
    const int64_t total = (int64_t)a->N * a->H * a->W * a->C;
    hipLaunchKernelGGL(dual_prior_kernel<0>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, *a, total);
    RET_LAUNCH();
}

extern "C" int dcvc_dual_prior_dec_index(const dcvc_dual_prior_args *a, void *stream) {
    if (!a || !a->fusion || !a->params || !a->idx || !a->idx_edges || (a->C & 1) || (a->step == 1 && !a->spatial))
        return DCVC_E_ARG;
    const int64_t total = (int64_t)a->N * a->H * a->W * a->C;
    hipLaunchKernelGGL(dual_prior_kernel<1>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, *a, total);
    RET_LAUNCH();
}

extern "C" int dcvc_dual_prior_dec_apply(const dcvc_dual_prior_args *a, void *stream) {
    if (dual_prior_check(a) || !a->sym) return DCVC_E_ARG;
    const int64_t total = (int64_t)a->N * a->H * a->W * a->C;
    hipLaunchKernelGGL(dual_prior_kernel<2>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, *a, total);
    RET_LAUNCH();
}

extern "C" int32_t dcvc_cdf_table_cols(void) { return kTabCols; }

extern "C" int dcvc_build_scale_cdfs(const float *scales, int32_t n_scales, int32_t kind, int32_t *cdf, int32_t *sizes,
                                     int32_t *offsets, void *stream) {
    if (!scales || !cdf || !sizes || !offsets || n_scales <= 0 || (kind != 0 && kind != 1)) return DCVC_E_ARG;
    hipLaunchKernelGGL(scale_cdfs_kernel, dim3(n_scales), dim3(128), 0, (hipStream_t)stream, scales, kind, cdf, sizes, offsets);
    RET_LAUNCH();
}

extern "C" int dcvc_build_factorized_cdfs(const float *params, int32_t C, int32_t *cdf, int32_t *sizes, int32_t *offsets,
                                          void *stream) {
    if (!params || !cdf || !sizes || !offsets || C <= 0 || C > 256) return DCVC_E_ARG;
    const size_t lds = (size_t)(2 * C + 4) * sizeof(int) + (size_t)C * kTabCols * sizeof(float);
    static bool attr_ok = hipFuncSetAttribute((const void *)factorized_cdfs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              160 * 1024) == hipSuccess;
    if (!attr_ok && lds > 64 * 1024) return DCVC_E_LAUNCH;
    hipLaunchKernelGGL(factorized_cdfs_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, params, C, cdf, sizes, offsets);
    RET_LAUNCH();
}

extern "C" int dcvc_scale_bits(const float *y_q, const float *scales_hat, float *out, float *scratch, int32_t kind,
                               int32_t N, int64_t per_sample, void *stream) {
    if (!y_q || !scales_hat || !out || !scratch) return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(scale_bits_kernel, dim3(RB, N), dim3(256), 0, st, y_q, scales_hat, scratch, kind, per_sample);
    hipLaunchKernelGGL(finish_sum, dim3(N), dim3(256), 0, st, scratch, out, RB);
    RET_LAUNCH();
}

extern "C" int dcvc_factorized_bits(const float *z_hat, int32_t z_cs, const float *params, float *out, float *scratch,
                                    int32_t N, int32_t HW, int32_t C, void *stream) {
    if (!z_hat || !params || !out || !scratch) return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(factorized_bits_kernel, dim3(RB, N), dim3(256), 0, st, z_hat, z_cs, params, scratch, (int64_t)HW,
                       C);
    hipLaunchKernelGGL(finish_sum, dim3(N), dim3(256), 0, st, scratch, out, RB);
    RET_LAUNCH();
}

extern "C" int dcvc_sq_err(const float *a, int32_t a_cs, const float *b, int32_t b_cs, float *out, float *scratch,
                           int32_t N, int32_t HW, int32_t C, void *stream) {
    if (!a || !b || !out || !scratch) return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sq_err_kernel, dim3(RB, N), dim3(256), 0, st, a, a_cs, b, b_cs, scratch, (int64_t)HW, C);
    hipLaunchKernelGGL(finish_sum, dim3(N), dim3(256), 0, st, scratch, out, RB);
    RET_LAUNCH();
}

extern "C" const char *dcvc_hip_version(void) { return "dcvc-hip 0.1 (gfx950, fp32 mfma)"; }
