// backward.hip -- gradient kernels of the DCVC-HEM P-frame path (include/dcvc_hip_grad.h).
//
// The reference gets these from torch.autograd when trainer.py calls loss.backward() on
// DMC.forward_one_frame's outputs (/root/reference/core/model/dcvc_hem.py:205-229,
// DCVC_HEM/src/models/video_model.py:470-596).  Here:
//   * the data gradient of a convolution is the forward MFMA kernel (conv_mfma.hip) run on the
//     flipped / channel-transposed filter, packed on the device by pack_kernel below;
//   * the weight gradient is its own MFMA kernel with the pixels as reduction dimension
//     (wgrad_kernel): C[co][ci] += dY[pixel][co] * X[pixel + tap][ci], fp32 32x32x2 MFMA,
//     K = 2 neighbouring pixels per instruction, one 32x32 accumulator per filter tap;
//   * everything else is HBM-bound elementwise / gather / scatter work fused per reference
//     function (epilogue backward, warp backward, dual-prior backward, likelihood backward).
// Reductions use fixed-order partial sums; only the scatter kernels (warp / up2 backward) use
// float atomics, exactly where ATen does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "dcvc_hip_grad.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

#define RET_LAUNCH() return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH
inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

constexpr int KC = 16;               // must match conv_mfma.hip
constexpr float WGT_SCALE = 64.f;    // must match conv_mfma.hip
constexpr float F16_MAX = 65504.f;

__device__ __forceinline__ float block_sum(float v, float *sm) {
    const int t = threadIdx.x;
    __syncthreads();
    sm[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) sm[t] += sm[t + s];
        __syncthreads();
    }
    return sm[0];
}

// ---------------------------------------------------------------------------------------------
// device-side weight packing
struct PackK {
    const float *w;
    const float *b;
    float *wpack;
    float *bpack;
    int Npk;        // output channels of the packed convolution
    int cp;         // padded
    int T, ks;
    int nseg;
    int seg_C[DCVC_MAX_SEG], seg_chunk0[DCVC_MAX_SEG], seg_cin0[DCVC_MAX_SEG];
    int CinT;       // Cin_total of the source tensor
    int cin_offset;
    int ps, precision, transposed;
    int64_t total;  // packed elements (chunks * T * 4 * cp * 4)
};

__device__ __forceinline__ void pack_body(const PackK &a, const int64_t gid) {
    if (gid < a.cp) {
        float bv = 0.f;
        if (!a.transposed && a.b && gid < a.Npk) {
            const int np = (int)gid;
            const int Cq = a.Npk >> 2;
            const int n = a.ps ? (np % Cq) * 4 + np / Cq : np;
            bv = a.b[n];
        }
        a.bpack[gid] = bv;
    }
    if (gid >= a.total) return;
    const int j = (int)(gid & 3);
    const int np = (int)((gid >> 2) % a.cp);
    const int kq = (int)((gid / (4 * (int64_t)a.cp)) & 3);
    const int t = (int)((gid / (16 * (int64_t)a.cp)) % a.T);
    const int cg = (int)(gid / (16 * (int64_t)a.cp * a.T));
    int s = 0;
    for (int i = 1; i < a.nseg; ++i)
        if (cg >= a.seg_chunk0[i]) s = i;
    const int c = (cg - a.seg_chunk0[s]) * KC + kq * 4 + j;
    float v = 0.f;
    if (c < a.seg_C[s] && np < a.Npk) {
        const int Cq = a.Npk >> 2;
        const int n = a.ps ? (np % Cq) * 4 + np / Cq : np;
        if (!a.transposed)
            v = a.w[((size_t)n * a.CinT + a.cin_offset + a.seg_cin0[s] + c) * a.T + t];
        else  // packed input channel c = forward output channel, packed output n = forward input channel
            v = a.w[((size_t)c * a.CinT + a.cin_offset + n) * a.T + (a.T - 1 - t)];
    }
    if (a.precision == DCVC_PREC_FP32) {
        a.wpack[gid] = v;
    } else {
        const int cc = kq * 4 + j, h = cc >> 3, jj = cc & 7;
        float sv = v * WGT_SCALE;
        sv = fminf(fmaxf(sv, -F16_MAX), F16_MAX);
        const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
        _Float16 *base = (_Float16 *)a.wpack;
        base[((((size_t)cg * a.T + t) * 4 + h) * a.cp + np) * 8 + jj] = hi;
        base[((((size_t)cg * a.T + t) * 4 + 2 + h) * a.cp + np) * 8 + jj] = lo;
    }
}

__global__ void pack_kernel(const PackK a) { pack_body(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x); }

// every layer of a plan in one launch: block b belongs to the entry e with first[e] <= b < first[e + 1]
__global__ void pack_batch_kernel(const PackK *__restrict__ table, const int *__restrict__ first, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    pack_body(table[lo], (int64_t)(blockIdx.x - first[lo]) * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// epilogue backward
__global__ void conv_bwd_prologue_kernel(const dcvc_conv_bwd_args a, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % a.Cout);
    const int64_t p = gid / a.Cout;
    const int ox = (int)(p % a.Wo), oy = (int)((p / a.Wo) % a.Ho);
    const int64_t n = p / ((int64_t)a.Wo * a.Ho);
    int cf = c, fy = oy, fx = ox, Hf = a.Ho, Wf = a.Wo, Cfin = a.Cout;
    if (a.pixel_shuffle) {  // nn.PixelShuffle(2): out[cf][2y+dy][2x+dx] = pre[cf*4 + dy*2 + dx][y][x]
        cf = c >> 2;
        fy = 2 * oy + ((c >> 1) & 1);
        fx = 2 * ox + (c & 1);
        Hf = 2 * a.Ho;
        Wf = 2 * a.Wo;
        Cfin = a.Cout >> 2;
    }
    const int64_t fp = (n * Hf + fy) * (int64_t)Wf + fx;
    float g = a.dout[fp * a.dout_cs + cf];
    float o = (a.act && a.out) ? a.out[fp * a.out_cs + cf] : 0.f;
    if (a.res2) {
        if (a.dres2) a.dres2[fp * a.dres2_cs + cf] += g;
        if (a.act) o -= a.res2[fp * a.res2_cs + cf];
    }
    if (a.res) {
        const float gt = a.gate ? a.gate[n * Cfin + cf] : 1.f;
        if (a.dres) a.dres[fp * a.dres_cs + cf] += g * gt;
        if (a.act) o -= a.res[fp * a.res_cs + cf] * gt;
    }
    if (a.act) g *= (o > 0.f) ? 1.f : a.slope;
    a.dpre[((n * a.Hd + (int64_t)oy * a.zs) * a.Wd + (int64_t)ox * a.zs) * a.dpre_cs + c] = g;
}

// ---------------------------------------------------------------------------------------------
// weight gradient
struct WgradK {
    const float *x;
    int x_cs, C, in_act;
    float in_slope;
    const float *dpre;
    int dpre_cs, zs, Hd, Wd;
    int N, Hin, Win, Ho, Wo, Cout;
    float *scratch;
    float *bscratch;  // per-split bias-gradient partials [splits][nco][32], or NULL
    int nci;        // 32-channel input tiles
    int ntx, nty;   // spatial tiles
    int T;
};

template <int KS, int S>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradK a) {
    constexpr int R = 4, TW = S == 1 ? 32 : 16;
    constexpr int KY = KS == 7 ? 1 : KS;  // filter rows per tap group (7x7: one row per blockIdx.z)
    constexpr int TG = KY * KS;
    constexpr int PH = (R - 1) * S + KY, PW = (TW - 1) * S + KS, PAD = KS / 2;
    constexpr int XS = PH * PW * 32 > 4096 ? PH * PW * 32 : 4096;
    __shared__ __attribute__((aligned(16))) float dys[R * TW * 32];
    __shared__ __attribute__((aligned(16))) float xs[XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hh = lane >> 5;
    const int cot = blockIdx.y / a.nci, cit = blockIdx.y % a.nci;
    const int co0 = cot * 32, ci0 = cit * 32;
    const int grp = blockIdx.z;  // filter row for 7x7, 0 otherwise
    const int ky0 = KS == 7 ? grp : 0;

    f32x16 acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // the workgroups of input tile 0 (and filter row 0 for 7x7) also sum dY over the pixels: the
    // bias gradient comes out of the same pass instead of a separate reduction over dY
    const bool do_bias = a.bscratch != nullptr && cit == 0 && grp == 0;
    float bsum = 0.f;

    // Software pipeline over the workgroup's tiles: the global loads of tile k+1 are in flight in
    // registers while the MFMAs of tile k run; channel masks and the activation are applied when the
    // registers are written to LDS, so nothing waits on the loads early.
    constexpr int ND = (R * TW * 8 + 255) / 256, NX = (PH * PW * 8 + 255) / 256;
    f32x4 rd[ND], rx[NX];
    unsigned okd = 0, okx = 0;
    const bool vd = !(a.dpre_cs & 3) && !((uintptr_t)a.dpre & 15);
    const bool vx = !(a.x_cs & 3) && !((uintptr_t)a.x & 15);
    auto load4 = [](const float *src, bool vec, int c, int C) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            v = *(const f32x4 *)src;  // in bounds: c + 3 < round4(C) <= channel stride
        } else {
            v[0] = src[0];
            if (c + 1 < C) v[1] = src[1];
            if (c + 2 < C) v[2] = src[2];
            if (c + 3 < C) v[3] = src[3];
        }
        return v;
    };
    auto load_tile = [&](int tile) {
        const int tx = tile % a.ntx, ty = (tile / a.ntx) % a.nty, n = tile / (a.ntx * a.nty);
        okd = 0;
        okx = 0;
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int i = tid + u * 256;
            const int p = i >> 3, q = i & 7;
            const int r = p / TW, xx = p - r * TW;
            const int oy = ty * R + r, ox = tx * TW + xx;
            const int c = co0 + q * 4;
            if (i < R * TW * 8 && oy < a.Ho && ox < a.Wo && c < a.Cout) {
                rd[u] = load4(a.dpre + (((size_t)n * a.Hd + (size_t)oy * a.zs) * a.Wd + (size_t)ox * a.zs) * a.dpre_cs + c, vd, c,
                              a.Cout);
                okd |= 1u << u;
            }
        }
        const int gy0 = ty * R * S - PAD + ky0, gx0 = tx * TW * S - PAD;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int i = tid + u * 256;
            const int p = i >> 3, q = i & 7;
            const int py = p / PW, px = p - py * PW;
            const int gy = gy0 + py, gx = gx0 + px;
            const int c = ci0 + q * 4;
            if (i < PH * PW * 8 && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win && c < a.C) {
                rx[u] = load4(a.x + (((size_t)n * a.Hin + gy) * a.Win + gx) * a.x_cs + c, vx, c, a.C);
                okx |= 1u << u;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int i = tid + u * 256;
            if (i < R * TW * 8) {
                const int c = co0 + (i & 7) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((okd >> u) & 1u) {
                    v = rd[u];
                    if (c + 1 >= a.Cout) v[1] = 0.f;
                    if (c + 2 >= a.Cout) v[2] = 0.f;
                    if (c + 3 >= a.Cout) v[3] = 0.f;
                }
                *(f32x4 *)&dys[(i >> 3) * 32 + (i & 7) * 4] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int i = tid + u * 256;
            if (i < PH * PW * 8) {
                const int c = ci0 + (i & 7) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((okx >> u) & 1u) {
                    v = rx[u];
                    if (c + 1 >= a.C) v[1] = 0.f;
                    if (c + 2 >= a.C) v[2] = 0.f;
                    if (c + 3 >= a.C) v[3] = 0.f;
                    if (a.in_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.in_slope;
                    }
                }
                *(f32x4 *)&xs[(i >> 3) * 32 + (i & 7) * 4] = v;
            }
        }
    };

    const int ntiles = a.N * a.nty * a.ntx;
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();  // the previous tile's LDS reads are done
        store_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        if (do_bias) {
#pragma unroll 4
            for (int p = tid >> 5; p < R * TW; p += 8) bsum += dys[p * 32 + (tid & 31)];
        }
        const int r = wave;  // one output row of the tile per wave
#pragma unroll 4
        for (int xp = 0; xp < TW; xp += 2) {
            const float av = dys[(r * TW + xp + hh) * 32 + col];
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int ky = t / KS, kx = t - ky * KS;
                const float bv = xs[((r * S + ky) * PW + (xp + hh) * S + kx) * 32 + col];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
    // cross-wave reduction, one tap at a time, then the block's partial goes to scratch
    const int nct = gridDim.y;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) xs[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col] = acc[t][r];
        __syncthreads();
        float *dst = a.scratch + (((size_t)blockIdx.x * nct + blockIdx.y) * a.T + (grp * TG + t)) * 1024;
        for (int i = tid; i < 1024; i += 256) dst[i] = (xs[i] + xs[1024 + i]) + (xs[2048 + i] + xs[3072 + i]);
    }
    if (do_bias) {
        __syncthreads();
        xs[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += xs[k * 32 + tid];
            a.bscratch[((size_t)blockIdx.x * (gridDim.y / a.nci) + cot) * 32 + tid] = s;
        }
    }
}

// The same weight gradient on the bf16 matrix cores (fast mode, stride 1): every fp32 operand is carried as
// hi = bf16(v), lo = bf16(v - hi) and dY.X ~= hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation --
// K = 16 pixels per instruction instead of 2, 3 x 32 cycles instead of 8 x 64 per 16 pixels.  bf16 keeps fp32's
// exponent, so gradients of any magnitude need no scaling; the dropped lo.lo term and the truncation of lo bound the
// error at ~2^-16 of sum |dY||X| per product (the fixtures' tolerance for gradients is 5e-3).  The MFMA wants 8
// CONSECUTIVE K (= pixels) per lane, i.e. both operands pixel-major: the staging pass transposes while it converts
// (4 pixels x 4 channels per thread -> one 8-byte LDS write per channel and plane).  A filter tap shifts the pixel run
// of X by kx elements; the shifted fragments are cut out of two aligned 16-byte reads with v_alignbit.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// (a, b) -> packed bf16 pair of the values and of their residuals
__device__ __forceinline__ void split_pair_bf16(float a, float b, unsigned &hi, unsigned &lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){ra, rb}, bf16x2));
}

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradK a) {
    constexpr int R = 4, TW = 32;
    constexpr int KY = KS == 7 ? 1 : KS, TG = KY * KS, PAD = KS / 2;
    constexpr int PH = R - 1 + KY, PW = TW + KS - 1;
    constexpr int XROW = 20;                                // dwords per (channel, patch row): 40 pixels >= PW
    constexpr int XCH = (PH * XROW + 63) / 64 * 64 + 4;     // channel stride = 4 (mod 64) dwords: 16-byte reads of
    constexpr int DCH = R * TW / 2 + 4;                     //   consecutive channels fall in distinct banks
    constexpr int NXG = PH * 10;                            // 4-pixel groups of the X patch
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * 32 * XCH + 2 * 32 * DCH];
    unsigned *xh = lds, *xl = lds + 32 * XCH, *dh = lds + 2 * 32 * XCH, *dl = dh + 32 * DCH;
    static_assert(sizeof(lds) >= 4096 * 4 && PW <= 40, "epilogue reuses the buffer for the cross-wave reduction");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hh = lane >> 5;
    const int cot = blockIdx.y / a.nci, cit = blockIdx.y % a.nci;
    const int co0 = cot * 32, ci0 = cit * 32;
    const int grp = blockIdx.z;
    const int ky0 = KS == 7 ? grp : 0;

    f32x16 acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias = a.bscratch != nullptr && cit == 0 && grp == 0;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};

    // staging items: (4-pixel group g, channel quad q).  dY: 32 groups x 8 quads = one per thread; X: NXG x 8, two per thread
    constexpr int NX = (NXG * 8 + 255) / 256;
    f32x4 rd[4], rx[NX][4];
    const bool vd = !(a.dpre_cs & 3) && !((uintptr_t)a.dpre & 15);
    const bool vx = !(a.x_cs & 3) && !((uintptr_t)a.x & 15);
    auto load4 = [](const float *src, bool vec, int c, int C) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            v = *(const f32x4 *)src;
        } else {
            v[0] = src[0];
            if (c + 1 < C) v[1] = src[1];
            if (c + 2 < C) v[2] = src[2];
            if (c + 3 < C) v[3] = src[3];
        }
        return v;
    };
    const int q = tid & 7;
    auto load_tile = [&](int tile) {
        const int tx = tile % a.ntx, ty = (tile / a.ntx) % a.nty, n = tile / (a.ntx * a.nty);
        {
            const int g = tid >> 3, r = g >> 3, x4 = g & 7;
            const int oy = ty * R + r, c = co0 + q * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = tx * TW + x4 * 4 + i;
                rd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (oy < a.Ho && ox < a.Wo && c < a.Cout)
                    rd[i] = load4(a.dpre + (((size_t)n * a.Hd + oy) * a.Wd + ox) * a.dpre_cs + c, vd, c, a.Cout);
            }
        }
        const int gy0 = ty * R - PAD + ky0, gx0 = tx * TW - PAD;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int g = (tid + u * 256) >> 3;
            const int py = g / 10, x4 = g - py * 10;
            const int gy = gy0 + py, c = ci0 + q * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = x4 * 4 + i, gx = gx0 + px;
                rx[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (g < NXG && px < PW && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win && c < a.C)
                    rx[u][i] = load4(a.x + (((size_t)n * a.Hin + gy) * a.Win + gx) * a.x_cs + c, vx, c, a.C);
            }
        }
    };
    auto store_tile = [&]() {
        {
            const int g = tid >> 3;
            const int c = co0 + q * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = c + e < a.Cout ? rd[i][e] : 0.f;
                if (do_bias) bs[e] += (v[0] + v[1]) + (v[2] + v[3]);
                unsigned h0, h1, l0, l1;
                split_pair_bf16(v[0], v[1], h0, l0);
                split_pair_bf16(v[2], v[3], h1, l1);
                *(u32x2 *)&dh[(q * 4 + e) * DCH + g * 2] = (u32x2){h0, h1};
                *(u32x2 *)&dl[(q * 4 + e) * DCH + g * 2] = (u32x2){l0, l1};
            }
        }
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int g = (tid + u * 256) >> 3;
            if (g < NXG) {
                const int py = g / 10, x4 = g - py * 10;
                const int c = ci0 + q * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = c + e < a.C ? rx[u][i][e] : 0.f;
                        if (a.in_act) v[i] = v[i] > 0.f ? v[i] : v[i] * a.in_slope;
                    }
                    unsigned h0, h1, l0, l1;
                    split_pair_bf16(v[0], v[1], h0, l0);
                    split_pair_bf16(v[2], v[3], h1, l1);
                    *(u32x2 *)&xh[(q * 4 + e) * XCH + py * XROW + x4 * 2] = (u32x2){h0, h1};
                    *(u32x2 *)&xl[(q * 4 + e) * XCH + py * XROW + x4 * 2] = (u32x2){l0, l1};
                }
            }
        }
    };

    const int ntiles = a.N * a.nty * a.ntx;
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        const int r = wave;  // one output row of the tile per wave; K = its 32 pixels in two steps of 16
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ao = col * DCH + r * 16 + ks * 8 + hh * 4;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4 *)&dh[ao]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4 *)&dl[ao]);
#pragma unroll
            for (int ky = 0; ky < KY; ++ky) {
                const int bo = col * XCH + (r + ky) * XROW + ks * 8 + hh * 4;
                unsigned wh[8], wl[8];
                *(u32x4 *)&wh[0] = *(const u32x4 *)&xh[bo];
                *(u32x4 *)&wl[0] = *(const u32x4 *)&xl[bo];
                if (KS > 1) {
                    *(u32x4 *)&wh[4] = *(const u32x4 *)&xh[bo + 4];
                    *(u32x4 *)&wl[4] = *(const u32x4 *)&xl[bo + 4];
                }
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    u32x4 fh, fl;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int w0 = d + kx / 2;
                        if (kx & 1) {
                            fh[d] = __builtin_amdgcn_alignbit(wh[w0 + 1], wh[w0], 16);
                            fl[d] = __builtin_amdgcn_alignbit(wl[w0 + 1], wl[w0], 16);
                        } else {
                            fh[d] = wh[w0];
                            fl[d] = wl[w0];
                        }
                    }
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, fh), bl = __builtin_bit_cast(bf16x8, fl);
                    const int t = ky * KS + kx;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // cross-wave reduction, one tap at a time, then the block's partial goes to scratch (layout of wgrad_kernel)
    float *xs = (float *)lds;
    const int nct = gridDim.y;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) xs[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col] = acc[t][r];
        __syncthreads();
        float *dst = a.scratch + (((size_t)blockIdx.x * nct + blockIdx.y) * a.T + (grp * TG + t)) * 1024;
        for (int i = tid; i < 1024; i += 256) dst[i] = (xs[i] + xs[1024 + i]) + (xs[2048 + i] + xs[3072 + i]);
    }
    if (do_bias) {  // thread (group g, quad q) holds the sums of channels 4q..4q+3 over its pixels
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[(tid >> 3) * 32 + q * 4 + e] = bs[e];
        __syncthreads();
        if (tid < 32) {
            float s = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) s += xs[k * 32 + tid];
            a.bscratch[((size_t)blockIdx.x * (gridDim.y / a.nci) + cot) * 32 + tid] = s;
        }
    }
}

// Threads walk the partials in their own order (tile, tap, 32x32 element) so the reads of every
// split are coalesced; four thread groups share the splits of an element.
__global__ void wgrad_reduce_kernel(const float *__restrict__ scratch, float *__restrict__ dw, int splits, int nct,
                                    int nci, int T, int Cout, int C, int CinT, int cin_offset, int overwrite,
                                    const float *__restrict__ bscratch, float *__restrict__ db, int nb_main) {
    __shared__ float sm[256];
    const int t = threadIdx.x, lane = t & 63, part = t >> 6;
    if ((int)blockIdx.x >= nb_main) {  // trailing workgroups: the bias gradient's partials
        const int nco = nct / nci;
        const int co = ((int)blockIdx.x - nb_main) * 64 + lane;
        float s = 0.f;
        if (co < nco * 32)
            for (int k = part; k < splits; k += 4) s += bscratch[(size_t)k * nco * 32 + co];
        sm[t] = s;
        __syncthreads();
        if (part == 0 && co < Cout) db[co] = (sm[lane] + sm[64 + lane]) + (sm[128 + lane] + sm[192 + lane]);
        return;
    }
    const int64_t idx = (int64_t)blockIdx.x * 64 + lane;  // over nct * T * 1024
    const int64_t total = (int64_t)nct * T * 1024;
    float s = 0.f;
    if (idx < total)
        for (int k = part; k < splits; k += 4) s += scratch[(size_t)k * total + idx];
    sm[t] = s;
    __syncthreads();
    if (part == 0 && idx < total) {
        const float r = (sm[lane] + sm[64 + lane]) + (sm[128 + lane] + sm[192 + lane]);
        const int e = (int)(idx & 1023);
        const int tap = (int)((idx >> 10) % T);
        const int tile = (int)(idx / ((int64_t)T * 1024));
        const int co = (tile / nci) * 32 + (e >> 5), ci = (tile % nci) * 32 + (e & 31);
        if (co < Cout && ci < C) {
            float *d = &dw[((size_t)co * CinT + cin_offset + ci) * T + tap];
            *d = overwrite ? r : *d + r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-channel dot products
__global__ void channel_dot_partial(const float *__restrict__ a, int a_cs, const float *__restrict__ b, int b_cs,
                                    float *__restrict__ scratch, int HW, int C, int NB) {
    __shared__ float sm[256];
    const int n = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
    const int cb = blockIdx.z * 256;
    const int Cl = min(C - cb, 256);
    const int G = 256 / Cl;
    const int cl = t % Cl, g = t / Cl;
    const int per = (HW + NB - 1) / NB;
    const int p0 = blk * per, p1 = min(HW, p0 + per);
    float v = 0.f;
    if (g < G)
        for (int p = p0 + g; p < p1; p += G) {
            const size_t pix = (size_t)n * HW + p;
            const float av = a[pix * a_cs + cb + cl];
            v += b ? av * b[pix * b_cs + cb + cl] : av;
        }
    sm[t] = v;
    __syncthreads();
    if (t < Cl) {
        float s = 0.f;
        for (int k = 0; k < G; ++k) s += sm[k * Cl + t];
        scratch[((size_t)n * NB + blk) * C + cb + t] = s;
    }
}

// grid (ceil(C/16), over_batch ? 1 : N): a block finishes 16 channels with 16 interleaved partial
// sums each, combined in a fixed order
__global__ void channel_dot_finish(const float *__restrict__ scratch, float *__restrict__ out, int N, int C, int NB,
                                   int over_batch, int accumulate) {
    __shared__ float sm[256];
    const int t = threadIdx.x;
    const int c = blockIdx.x * 16 + (t & 15), part = t >> 4;
    const int n0 = over_batch ? 0 : blockIdx.y, n1 = over_batch ? N : n0 + 1;
    float s = 0.f;
    if (c < C)
        for (int n = n0; n < n1; ++n)
            for (int k = part; k < NB; k += 16) s += scratch[((size_t)n * NB + k) * C + c];
    sm[t] = s;
    __syncthreads();
    if (t < 16 && c < C) {
        float r = 0.f;
        for (int k = 0; k < 16; ++k) r += sm[k * 16 + t];
        float *o = out + (over_batch ? 0 : (size_t)n0 * C) + c;
        *o = accumulate ? *o + r : r;
    }
}

// ---------------------------------------------------------------------------------------------
// elementwise
__global__ void mask_accumulate_kernel(const float *__restrict__ src, int src_cs, const float *__restrict__ x, int x_cs,
                                       float slope, float *__restrict__ dst, int dst_cs, int64_t npix, int C) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * C) return;
    const int c = (int)(gid % C);
    const int64_t p = gid / C;
    const float v = src[p * src_cs + c];
    // (an explicit fma: the same single rounding as the fused form of this operation in the data-gradient convolution's
    // epilogue, dcvc_conv_args.out_act 3 with a residual -- whatever the compiler's contraction choices in either kernel)
    const float m = x ? (x[p * x_cs + c] > 0.f ? 1.f : slope) : 1.f;
    dst[p * dst_cs + c] = __builtin_fmaf(v, m, dst[p * dst_cs + c]);
}

__global__ void add_planes_kernel(const float *__restrict__ a, int a_cs, const float *__restrict__ b, int b_cs,
                                  float *__restrict__ out, int out_cs, int64_t npix, int C) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * C) return;
    const int c = (int)(gid % C);
    const int64_t p = gid / C;
    out[p * out_cs + c] = a[p * a_cs + c] + b[p * b_cs + c];
}

__global__ void add_channel_vec_kernel(float *__restrict__ dst, int dst_cs, const float *__restrict__ vec, float scale,
                                       int64_t HW, int C, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t p = gid / C;
    const int64_t n = p / HW;
    dst[p * dst_cs + c] += vec[n * C + c] * scale;
}

// ---------------------------------------------------------------------------------------------
// resampling backward.  Coordinates are rebuilt exactly as in resample.hip (make_tap).
__device__ __forceinline__ float lin11(int i, int n) {
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// Order-independent accumulation for the scatter of grid_sample's backward: contributions are scaled by a power of
// two, rounded to integers and added with 64-bit integer atomics (associative: run-to-run identical).  The scale is
// chosen PER CALL from max |dout| (an absolute 2^-36 quantum rounds feature-level gradients of 1e-9 at
// 1e-2 relative): 2^40 / 2^ceil(log2 max), i.e. the largest contribution lands near 2^40 -- 2^21 of them (every
// pixel of a 1080p picture sampling one source element) still fit 63 bits -- and the quantum is 2^-40 of the largest
// upstream gradient whatever its magnitude.  A non-finite upstream gradient poisons the whole result with NaN instead
// of being clamped to a finite integer.  The call's control word (fix[npix * C]): max |dout| bit pattern.
__device__ __forceinline__ int fix_scale_exp(unsigned maxbits) {  // exponent se of the call's scale 2^se
    const int e = (int)(maxbits >> 23) - 127;                      // floor(log2 max) for normal floats
    return 39 - (e < -80 ? -80 : e);                               // (<= 119: the scale itself stays a finite float)
}
__device__ __forceinline__ void fix_add(unsigned long long *p, float v, float scale) {
    atomicAdd(p, (unsigned long long)__float2ll_rn(v * scale));
}

__global__ void absmax_bits_kernel(const float *__restrict__ x, int x_cs, int64_t npix, int C, unsigned *__restrict__ word) {
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * C; i += (int64_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(x[(i / C) * x_cs + i % C]) & 0x7fffffffu);  // |x| orders like its bits; Inf / NaN on top
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(word, m);  // integer max: order-independent
}

__global__ void fix_finish_kernel(unsigned long long *__restrict__ fix, float *__restrict__ dst, int dst_cs, int64_t npix, int C) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * C) return;
    const unsigned maxbits = (unsigned)fix[npix * C];
    const long long v = (long long)fix[gid];
    fix[gid] = 0;  // left clean for the next call
    if (maxbits >= 0x7f800000u) {
        dst[(gid / C) * dst_cs + gid % C] = __uint_as_float(0x7fc00000u);  // upstream gradient was not finite
    } else if (v) {
        dst[(gid / C) * dst_cs + gid % C] += (float)ldexp((double)v, -fix_scale_exp(maxbits));
    }
}

__global__ void fix_reset_kernel(unsigned long long *word) { *word = 0; }

// One thread per (pixel, channel group): scatter into dsrc with integer atomics, reduce the flow gradient over the Cl
// lanes of a pixel (consecutive lanes of one wave) with a shuffle butterfly.  Round 2: this reduction used to go through
// LDS (write, barrier, lane 0 sums); with the bf16 weight-gradient kernels running on the second stream its x component
// came out different in a few pixels from run to run (tools/repro_check.py at batch 4 x 256x256: 134 parameter
// gradients).  Reproduced in isolation by tools/warp_bwd_lds_probe.py (profiles/r02_warp_bwd_lds_probe.txt): every LDS
// formulation tried (arrays padded, swapped, volatile, extra register copies; with or without the scatter's atomics in
// the loop) differs run to run in the x component of a few pixels while bf16-MFMA kernels run on another stream, and
// never alone or beside fp32-MFMA kernels; this shuffle version never differs -- and neither does the LDS version once
// it is compiled without packed-FP32 instructions (-fno-slp-vectorize: no v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32):
// the value that went wrong was the LOW half of the (gx, gy) pair the compiler carried through chains of packed
// operations.  Whether a wait state is missing in that code or the hardware misbehaves when packed-FP32 work shares a
// SIMD with 16-bit MFMAs of another kernel is open; the library is therefore built without the packed-fp32 target
// feature (csrc/Makefile), and tests/test_gpu_backward.py::test_training_step_is_bit_reproducible guards four sizes.
__global__ void warp_bwd_kernel(const float *__restrict__ src, int src_cs, const float *__restrict__ flow, int flow_cs,
                                const float *__restrict__ dout, int dout_cs, float *__restrict__ dsrc, int dsrc_cs,
                                float *__restrict__ dflow, int dflow_cs, int N, int H, int W, int C,
                                unsigned long long *__restrict__ fix) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.y + threadIdx.y;  // blockDim = (Cl, 256 / Cl)
    const bool live = pix < (int64_t)N * H * W;
    float gx_acc = 0.f, gy_acc = 0.f;
    float fscale = 0.f;  // this call's fixed-point scale (0: a non-finite upstream gradient, fix_finish writes NaN)
    if (dsrc) {
        const unsigned maxbits = (unsigned)fix[(int64_t)N * H * W * C];
        if (maxbits < 0x7f800000u) fscale = ldexpf(1.f, fix_scale_exp(maxbits));
    }
    if (live) {
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const int64_t n = pix / ((int64_t)W * H);
        const float fx = flow[pix * flow_cs], fy = flow[pix * flow_cs + 1];
        const float hx = (float)(((double)W - 1.0) / 2.0), hy = (float)(((double)H - 1.0) / 2.0);
        float ix = (lin11(x, W) + fx / hx + 1.0f) * hx, iy = (lin11(y, H) + fy / hy + 1.0f) * hy;
        // clip_coordinates_set_grad: zero gradient at and beyond the border
        const float mx = (ix <= 0.f || ix >= (float)(W - 1)) ? 0.f : 1.f;
        const float my = (iy <= 0.f || iy >= (float)(H - 1)) ? 0.f : 1.f;
        ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
        iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
        const float xw = floorf(ix), yn = floorf(iy);
        const float w = ix - xw, e = 1.0f - w, nn = iy - yn, s = 1.0f - nn;
        const int x0 = (int)xw, y0 = (int)yn, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const bool x1in = x0 + 1 <= W - 1, y1in = y0 + 1 <= H - 1;  // ATen drops out-of-range taps
        const size_t b = (size_t)n * H * W;
        const size_t pnw = b + (size_t)y0 * W + x0, pne = b + (size_t)y0 * W + x1;
        const size_t psw = b + (size_t)y1 * W + x0, pse = b + (size_t)y1 * W + x1;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float g = dout[pix * dout_cs + c];
            const float vnw = src[pnw * src_cs + c];
            const float vne = x1in ? src[pne * src_cs + c] : 0.f;
            const float vsw = y1in ? src[psw * src_cs + c] : 0.f;
            const float vse = (x1in && y1in) ? src[pse * src_cs + c] : 0.f;
            gx_acc += g * (s * (vne - vnw) + nn * (vse - vsw));
            gy_acc += g * (e * (vsw - vnw) + w * (vse - vne));
            if (dsrc) {
                // the scatter is data dependent (several output pixels may sample one source pixel): summed as
                // 64-bit fixed point, whose addition is associative, so the result does not depend on the order
                // in which the atomics land (float atomics made two runs of the same step differ in the last bits)
                fix_add(&fix[pnw * C + c], g * (s * e), fscale);
                if (x1in) fix_add(&fix[pne * C + c], g * (s * w), fscale);
                if (y1in) fix_add(&fix[psw * C + c], g * (nn * e), fscale);
                if (x1in && y1in) fix_add(&fix[pse * C + c], g * (nn * w), fscale);
            }
        }
        gx_acc *= mx;
        gy_acc *= my;
    }
    if (!dflow) return;
    // the Cl (<= 64, a power of two) lanes of a pixel are consecutive lanes of one wave: butterfly sum in a fixed order
    for (int m = (int)blockDim.x >> 1; m >= 1; m >>= 1) {
        gx_acc += __shfl_xor(gx_acc, m);
        gy_acc += __shfl_xor(gy_acc, m);
    }
    if (threadIdx.x == 0 && live) {
        dflow[pix * dflow_cs] += gx_acc;
        dflow[pix * dflow_cs + 1] += gy_acc;
    }
}

// adjoint of up2_kernel as a GATHER: one thread per source element visits the <= 6 x 6 output pixels whose
// bilinear taps can touch it, recomputes their taps with the forward kernel's arithmetic and adds the matching
// weights in a fixed order (no atomics: run-to-run identical)
__global__ void up2_bwd_kernel(const float *__restrict__ dout, int dout_cs, float *__restrict__ dsrc, int dsrc_cs, int N,
                               int H, int W, int C, float scale) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)N * H * W * C) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int sx = (int)(pix % W), sy = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    const int Ho = 2 * H, Wo = 2 * W;
    auto weight = [](int o, int s, int L) {  // weight of source index s in output index o (0 if not a tap)
        const float p = fmaxf(((float)o + 0.5f) * 0.5f - 0.5f, 0.f);
        const int i0 = (int)p, i1 = min(i0 + 1, L - 1);
        const float l1 = p - (float)i0, l0 = 1.f - l1;
        return (i0 == s ? l0 : 0.f) + (i1 == s ? l1 : 0.f);
    };
    const float *b = dout + n * (int64_t)Ho * Wo * dout_cs + c;
    float acc = 0.f;
    for (int oy = max(0, 2 * sy - 2); oy <= min(Ho - 1, 2 * sy + 3); ++oy) {
        const float wy = weight(oy, sy, H);
        if (wy == 0.f) continue;
        for (int ox = max(0, 2 * sx - 2); ox <= min(Wo - 1, 2 * sx + 3); ++ox) {
            const float wx = weight(ox, sx, W);
            if (wx != 0.f) acc += (b[((int64_t)oy * Wo + ox) * dout_cs] * scale) * wy * wx;
        }
    }
    dsrc[pix * dsrc_cs + c] += acc;
}

__global__ void down2_bwd_kernel(const float *__restrict__ dout, int dout_cs, float *__restrict__ dsrc, int dsrc_cs,
                                 int N, int H, int W, int C, float k) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)N * H * W * C) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    const int64_t op = (n * (H / 2) + y / 2) * (int64_t)(W / 2) + x / 2;
    dsrc[pix * dsrc_cs + c] += dout[op * dout_cs + c] * k;
}

__global__ void maxpool2_bwd_kernel(const float *__restrict__ src, int src_cs, const float *__restrict__ dout,
                                    int dout_cs, float *__restrict__ dsrc, int dsrc_cs, int N, int H, int W, int C) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ho = H / 2, Wo = W / 2;
    if (gid >= (int64_t)N * Ho * Wo * C) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
    const int64_t n = pix / ((int64_t)Wo * Ho);
    const int64_t p00 = (n * H + 2 * oy) * (int64_t)W + 2 * ox;
    const int64_t cand[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    int best = 0;
    float bv = src[cand[0] * src_cs + c];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const float v = src[cand[k] * src_cs + c];
        if (v > bv || v != v) {
            bv = v;
            best = k;
        }
    }
    dsrc[cand[best] * dsrc_cs + c] += dout[pix * dout_cs + c];
}

// ---------------------------------------------------------------------------------------------
// SE layer FC part: one block, all samples
__global__ void se_bwd_kernel(const float *__restrict__ mean, const float *__restrict__ w1, const float *__restrict__ w2,
                              const float *__restrict__ gate, const float *__restrict__ dgate, float *__restrict__ dmean,
                              float *__restrict__ dw1, float *__restrict__ dw2, int N, int C, int Cr) {
    __shared__ float hid[64], dhid[64], ds[256];
    const int t = threadIdx.x;
    for (int n = 0; n < N; ++n) {
        __syncthreads();
        if (t < Cr) {
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += w1[t * C + c] * mean[(size_t)n * C + c];
            hid[t] = fmaxf(s, 0.f);
        }
        if (t < C) {
            const float g = gate[(size_t)n * C + t];
            ds[t] = dgate[(size_t)n * C + t] * g * (1.f - g);
        }
        __syncthreads();
        if (t < Cr) {
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += w2[c * Cr + t] * ds[c];
            dhid[t] = hid[t] > 0.f ? s : 0.f;
        }
        if (t < C && dw2)
            for (int j = 0; j < Cr; ++j) dw2[t * Cr + j] += ds[t] * hid[j];
        __syncthreads();
        if (t < C) {
            float s = 0.f;
            for (int j = 0; j < Cr; ++j) s += w1[j * C + t] * dhid[j];
            dmean[(size_t)n * C + t] = s;
            if (dw1)
                for (int j = 0; j < Cr; ++j) dw1[j * C + t] += dhid[j] * mean[(size_t)n * C + t];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// q scales
__global__ void scale_channels_bwd_kernel(const float *__restrict__ dout, int dout_cs, float *__restrict__ dsrc,
                                          int dsrc_cs, const float *__restrict__ q_basic,
                                          const float *__restrict__ q_scale, int mode, int64_t HW, int C, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int n = (int)(pix / HW);
    const float q = fmaxf(q_basic[c], 0.5f) * q_scale[n];
    const float g = dout[pix * dout_cs + c];
    dsrc[pix * dsrc_cs + c] += mode ? g * q : g / q;
}

__global__ void q_finish_kernel(const float *__restrict__ dq_mul, const float *__restrict__ s_div,
                                const float *__restrict__ q_basic, const float *__restrict__ q_scale,
                                float *__restrict__ dq_basic, float *__restrict__ dq_scale, int N, int C) {
    __shared__ float sm[256];
    const int t = threadIdx.x;
    // per-channel: sum over samples
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + t;
        if (c < C && dq_basic) {
            const float qb = fmaxf(q_basic[c], 0.5f);
            float s = 0.f;
            for (int n = 0; n < N; ++n) {
                float d = dq_mul ? dq_mul[(size_t)n * C + c] : 0.f;
                if (s_div) d -= s_div[(size_t)n * C + c] / (qb * q_scale[n]);
                s += d * q_scale[n];
            }
            if (q_basic[c] >= 0.5f || s < 0.f) dq_basic[c] += s;
        }
    }
    if (!dq_scale) return;
    for (int n = 0; n < N; ++n) {
        float v = 0.f;
        for (int c = t; c < C; c += 256) {
            const float qb = fmaxf(q_basic[c], 0.5f);
            float d = dq_mul ? dq_mul[(size_t)n * C + c] : 0.f;
            if (s_div) d -= s_div[(size_t)n * C + c] / (qb * q_scale[n]);
            v += d * qb;
        }
        const float s = block_sum(v, sm);
        if (t == 0) dq_scale[n] += s;
    }
}

// ---------------------------------------------------------------------------------------------
// dual prior
__global__ void dual_prior_bwd_kernel(const dcvc_dual_prior_bwd_args a, int64_t total) {
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
    const bool active0 = half ? !m0 : m0;
    const int64_t e = pix * C + c;
    const float dres = a.dy_res ? a.dy_res[e] : 0.f;
    const float dsh = a.dscales_hat ? a.dscales_hat[e] : 0.f;
    if (a.step == 1) {
        float *ds = a.dspatial + pix * a.dspatial_cs;
        ds[half ? C + k : k] = active0 ? 0.f : dsh;
        ds[half ? C + Ch + k : Ch + k] = active0 ? 0.f : -dres;
        return;
    }
    const float *fu = a.fusion + pix * a.fusion_cs;
    const float qs = fmaxf(fu[c], 0.5f);
    const float cq = fmaxf(a.q_basic[c], 0.5f) * a.q_scale[n];
    const float hat = a.y_hat[e];
    const float g_out = a.dout ? a.dout[pix * a.dout_cs + c] : 0.f;
    const float *dp = a.dparams ? a.dparams + pix * a.dparams_cs : nullptr;
    float g_hat = g_out * qs * cq;
    if (active0 && dp) g_hat += dp[c];
    const float g_yq = g_hat + dres;
    float g_sc = active0 ? dsh : 0.f, g_mu = active0 ? -dres : 0.f, g_qs = 0.f;
    if (dp) {
        g_mu += dp[C + c];
        g_sc += dp[2 * C + c];
        g_qs += dp[3 * C + c];
    }
    const float yv = a.y[pix * a.y_cs + c];
    const float yq = yv / qs;
    g_qs += g_out * hat * cq - g_yq * yq / qs;
    float *df = a.dfusion + pix * a.dfusion_cs;
    if (fu[c] >= 0.5f || g_qs < 0.f) df[c] += g_qs;
    df[C + c] += g_sc;
    df[2 * C + c] += g_mu;
    if (a.dy) a.dy[pix * a.dy_cs + c] += g_yq / qs;
    if (a.dq_plane) a.dq_plane[e] = g_out * hat * qs;
}

// ---------------------------------------------------------------------------------------------
// likelihoods
__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

__global__ void scale_bits_bwd_kernel(const float *__restrict__ yv, const float *__restrict__ sh, const float *__restrict__ g,
                                      float *__restrict__ dy, float *__restrict__ dsc, int kind, int64_t per, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float up = g[i / per];
    const float y = yv[i], s_raw = sh[i];
    if (kind == 1) {
        // get_y_gaussian_bits (common_model.py:57-62; IntraNoAR): sigma.clamp(0.11, 1e10), Normal(0, sigma).cdf(v) =
        // 0.5 * (1 + erf(v * (1 / sigma) / sqrt(2))) as torch.distributions writes it; autograd: erf'(u) = 2/sqrt(pi) exp(-u^2),
        // (1 / sigma)' = -1 / sigma^2, clamp passes the gradient inside its range
        const float sg = fminf(fmaxf(s_raw, 0.11f), 1e10f);
        const float rs = 1.f / sg;
        const float t1 = y + 0.5f, t0 = y - 0.5f;
        const float u1 = t1 * rs / 1.4142135623730951f, u0 = t0 * rs / 1.4142135623730951f;
        const float p = 0.5f * (1.f + erff(u1)) - 0.5f * (1.f + erff(u0));
        const float bits = (-1.0f * logf(p + 1e-5f)) / 0.6931471805599453f;
        const float gb = (bits >= 0.f || up < 0.f) ? up : 0.f;
        const float gp = -gb / ((p + 1e-5f) * 0.6931471805599453f);
        const float c = 0.5f * 1.1283791670955126f;  // 0.5 * 2 / sqrt(pi)
        const float d1 = c * expf(-u1 * u1), d0 = c * expf(-u0 * u0);  // d cdf / d u
        dy[i] = gp * (d1 - d0) * rs / 1.4142135623730951f;
        const float du_ds1 = -t1 * rs * rs / 1.4142135623730951f, du_ds0 = -t0 * rs * rs / 1.4142135623730951f;
        const bool in_range = s_raw >= 0.11f && s_raw <= 1e10f;
        dsc[i] = in_range ? gp * (d1 * du_ds1 - d0 * du_ds0) : 0.f;
        return;
    }
    const float b = fminf(fmaxf(s_raw, 1e-5f), 1e10f);
    const float t1 = y + 0.5f, t0 = y - 0.5f;
    const float s1 = sgn(t1), s0 = sgn(t0);
    // torch.autograd differentiates expm1 as grad * (result + 1) (derivatives.yaml), i.e. with exp(u) REBUILT from the
    // fp32 value of expm1(u): for u << 0 that value sits on the 6e-8 grid next to -1 and the rebuilt exponential is a
    // multiple of 6e-8 (zero below 3e-8) instead of the tiny true one.  With bits = -log2(p + 1e-5) the factor in front is
    // up to 1.4e5, so those tails matter: the analytic exp(u) used here until round 3 made the rate gradients of the
    // hyper-prior path 2-8 % LARGER than the reference's (tests/diag/forced_grad_probe.py, same symbols on both sides).
    // The reference's arithmetic is the contract, so the kernel rebuilds the exponential the same way.
    const float m1 = expm1f(-fabsf(t1) / b), m0 = expm1f(-fabsf(t0) / b);
    const float e1 = m1 + 1.0f, e0 = m0 + 1.0f;
    const float F1 = 0.5f - 0.5f * s1 * m1, F0 = 0.5f - 0.5f * s0 * m0;
    const float p = F1 - F0;
    const float bits = (-1.0f * logf(p + 1e-5f)) / 0.6931471805599453f;
    float gb = (bits >= 0.f || up < 0.f) ? up : 0.f;          // LowerBound(bits, 0) backward
    const float gp = -gb / ((p + 1e-5f) * 0.6931471805599453f);
    // dF/dt = 0.5 sign(t)^2 exp(-|t|/b) / b ;  dF/db = -0.5 sign(t) exp(-|t|/b) |t| / b^2
    const float dFdt1 = 0.5f * s1 * s1 * e1 / b, dFdt0 = 0.5f * s0 * s0 * e0 / b;
    const float dFdb1 = -0.5f * s1 * e1 * fabsf(t1) / (b * b), dFdb0 = -0.5f * s0 * e0 * fabsf(t0) / (b * b);
    dy[i] = gp * (dFdt1 - dFdt0);
    const bool inside = s_raw >= 1e-5f && s_raw <= 1e10f;      // clamp passes the gradient inside its range
    dsc[i] = inside ? gp * (dFdb1 - dFdb0) : 0.f;
}

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// forward of the 4-layer per-channel CDF keeping what the backward needs; returns sigmoid(logit)
struct FactState {
    float xin[4], x1[3];  // layer inputs, pre-tanh values
};
__device__ __forceinline__ float fact_fwd(float x, const float *P, FactState &st) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        st.xin[i] = x;
        const float v = x * softplusf(P[3 * i]) + P[3 * i + 1];
        st.x1[i] = v;
        x = v + tanhf(v) * tanhf(P[3 * i + 2]);
    }
    st.xin[3] = x;
    x = x * softplusf(P[9]) + P[10];
    return sigmoidf(x);
}
// adds d(cdf)/d(params) * gcdf into dP[11] and returns d(cdf)/dx * gcdf
__device__ __forceinline__ float fact_bwd(float cdf, float gcdf, const float *P, const FactState &st, float *dP) {
    float gx = gcdf * cdf * (1.f - cdf);  // through the sigmoid
    dP[10] += gx;
    dP[9] += gx * st.xin[3] * sigmoidf(P[9]);
    gx *= softplusf(P[9]);
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        const float th = tanhf(st.x1[i]), ta = tanhf(P[3 * i + 2]);
        dP[3 * i + 2] += gx * th * (1.f - ta * ta);
        const float gv = gx * (1.f + (1.f - th * th) * ta);
        dP[3 * i + 1] += gv;
        dP[3 * i] += gv * st.xin[i] * sigmoidf(P[3 * i]);
        gx = gv * softplusf(P[3 * i]);
    }
    return gx;
}

// one block per channel; loops over all samples / pixels of the channel
__global__ void factorized_bits_bwd_kernel(const float *__restrict__ z, int z_cs, const float *__restrict__ P,
                                           const float *__restrict__ g, float *__restrict__ dz, int dz_cs,
                                           float *__restrict__ dparams, int N, int64_t HW, int C) {
    __shared__ float sm[256];
    const int c = blockIdx.x;
    float Pc[11], dP[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        Pc[i] = P[i * C + c];
        dP[i] = 0.f;
    }
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const float up = g[i / HW];
        const float zz = z[i * z_cs + c];
        FactState s1, s0;
        const float c1 = fact_fwd(zz + 0.5f, Pc, s1), c0 = fact_fwd(zz - 0.5f, Pc, s0);
        const float p = c1 - c0;
        const float bits = (-1.0f * logf(p + 1e-5f)) / 0.6931471805599453f;
        const float gb = (bits >= 0.f || up < 0.f) ? up : 0.f;
        const float gp = -gb / ((p + 1e-5f) * 0.6931471805599453f);
        const float gz = fact_bwd(c1, gp, Pc, s1, dP) + fact_bwd(c0, -gp, Pc, s0, dP);
        if (dz) dz[i * dz_cs + c] += gz;
    }
    if (!dparams) return;
    for (int i = 0; i < 11; ++i) {
        const float s = block_sum(dP[i], sm);
        if (threadIdx.x == 0) dparams[i * C + c] += s;
    }
}

__global__ void sq_err_bwd_kernel(const float *__restrict__ a, int a_cs, const float *__restrict__ b, int b_cs,
                                  const float *__restrict__ g, float *__restrict__ da, int da_cs, int64_t HW, int C,
                                  int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int64_t n = pix / HW;
    da[pix * da_cs + c] += 2.f * (a[pix * a_cs + c] - b[pix * b_cs + c]) * g[n];
}

}  // namespace

// =============================================================================================
static int make_pack(PackK &k, const float *w, const float *b, int32_t Cout, int32_t Cin_total, int32_t ks, int32_t nseg,
                     const int32_t *seg_C, int32_t cin_offset, int32_t pixel_shuffle, int32_t precision,
                     int32_t transposed, float *wpack, float *bpack) {
    if (!w || !wpack || !bpack || !seg_C || nseg < 1 || nseg > DCVC_MAX_SEG || (ks != 1 && ks != 3 && ks != 7))
        return DCVC_E_ARG;
    if (precision != DCVC_PREC_FP32 && precision != DCVC_PREC_FP16X3) return DCVC_E_ARG;
    if (transposed && (nseg != 1 || pixel_shuffle)) return DCVC_E_ARG;
    memset(&k, 0, sizeof(k));
    k.w = w;
    k.b = b;
    k.wpack = wpack;
    k.bpack = bpack;
    k.T = ks * ks;
    k.ks = ks;
    k.CinT = Cin_total;
    k.cin_offset = cin_offset;
    k.ps = pixel_shuffle;
    k.precision = precision;
    k.transposed = transposed;
    int chunks = 0, cin0 = 0;
    if (!transposed) {
        k.Npk = Cout;
        k.nseg = nseg;
        for (int s = 0; s < nseg; ++s) {
            if (seg_C[s] <= 0) return DCVC_E_ARG;
            k.seg_C[s] = seg_C[s];
            k.seg_chunk0[s] = chunks;
            k.seg_cin0[s] = cin0;
            chunks += (seg_C[s] + KC - 1) / KC;
            cin0 += seg_C[s];
        }
        if (cin_offset < 0 || cin_offset + cin0 > Cin_total) return DCVC_E_ARG;
    } else {
        if (seg_C[0] <= 0 || cin_offset < 0 || cin_offset + seg_C[0] > Cin_total) return DCVC_E_ARG;
        k.Npk = seg_C[0];
        k.nseg = 1;
        k.seg_C[0] = Cout;
        chunks = (Cout + KC - 1) / KC;
    }
    if (pixel_shuffle && (k.Npk & 3)) return DCVC_E_ARG;
    k.cp = round_up(k.Npk, 32);
    k.total = (int64_t)chunks * k.T * 4 * k.cp * 4;
    return DCVC_OK;
}

extern "C" int dcvc_conv_pack_weights_dev(const float *w, const float *b, int32_t Cout, int32_t Cin_total, int32_t ks,
                                          int32_t nseg, const int32_t *seg_C, int32_t cin_offset, int32_t pixel_shuffle,
                                          int32_t precision, int32_t transposed, float *wpack, float *bpack,
                                          void *stream) {
    PackK k;
    const int rc = make_pack(k, w, b, Cout, Cin_total, ks, nseg, seg_C, cin_offset, pixel_shuffle, precision, transposed,
                             wpack, bpack);
    if (rc != DCVC_OK) return rc;
    hipLaunchKernelGGL(pack_kernel, dim3(nblk(k.total, 256)), dim3(256), 0, (hipStream_t)stream, k);
    RET_LAUNCH();
}

// A plan = the packing jobs of every layer of a model, resident on the device: one launch re-packs them all after an
// optimiser step (a training step otherwise spends ~370 launches of a few microseconds each on this).
struct PackPlan {
    PackK *table;
    int *first;
    int n;
    unsigned blocks;
};

extern "C" int dcvc_pack_plan_create(const dcvc_pack_job *jobs, int32_t n, void **plan) {
    if (!jobs || n < 1 || !plan) return DCVC_E_ARG;
    std::vector<PackK> table((size_t)n);
    std::vector<int> first((size_t)n);
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        const dcvc_pack_job &j = jobs[i];
        const int rc = make_pack(table[i], j.w, j.b, j.Cout, j.Cin_total, j.ks, j.nseg, j.seg_C, j.cin_offset,
                                 j.pixel_shuffle, j.precision, j.transposed, j.wpack, j.bpack);
        if (rc != DCVC_OK) return rc;
        first[i] = (int)blocks;
        blocks += nblk(table[i].total > table[i].cp ? table[i].total : table[i].cp, 256);
        if (blocks > 0x7fffffff) return DCVC_E_ARG;
    }
    PackPlan *p = new PackPlan{nullptr, nullptr, n, (unsigned)blocks};
    if (hipMalloc((void **)&p->table, sizeof(PackK) * (size_t)n) != hipSuccess ||
        hipMalloc((void **)&p->first, sizeof(int) * (size_t)n) != hipSuccess ||
        hipMemcpy(p->table, table.data(), sizeof(PackK) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->first, first.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) {
        if (p->table) (void)hipFree(p->table);
        if (p->first) (void)hipFree(p->first);
        delete p;
        return DCVC_E_LAUNCH;
    }
    *plan = p;
    return DCVC_OK;
}

extern "C" int dcvc_pack_plan_run(void *plan, void *stream) {
    PackPlan *p = (PackPlan *)plan;
    if (!p || !p->table) return DCVC_E_ARG;
    hipLaunchKernelGGL(pack_batch_kernel, dim3(p->blocks), dim3(256), 0, (hipStream_t)stream, p->table, p->first, p->n);
    RET_LAUNCH();
}

extern "C" void dcvc_pack_plan_destroy(void *plan) {
    PackPlan *p = (PackPlan *)plan;
    if (!p) return;
    (void)hipFree(p->table);
    (void)hipFree(p->first);
    delete p;
}

extern "C" int dcvc_conv_bwd_prologue(const dcvc_conv_bwd_args *a, void *stream) {
    if (!a || !a->dout || !a->dpre || a->N <= 0 || a->Ho <= 0 || a->Wo <= 0 || a->Cout <= 0 || a->zs < 1) return DCVC_E_ARG;
    if (a->act && !a->out) return DCVC_E_ARG;
    if (a->pixel_shuffle && (a->Cout & 3)) return DCVC_E_ARG;
    if ((a->Ho - 1) * a->zs >= a->Hd || (a->Wo - 1) * a->zs >= a->Wd || a->dpre_cs < a->Cout) return DCVC_E_ARG;
    const int64_t total = (int64_t)a->N * a->Ho * a->Wo * a->Cout;
    hipLaunchKernelGGL(conv_bwd_prologue_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, *a, total);
    RET_LAUNCH();
}

extern "C" int64_t dcvc_conv_wgrad_scratch_min(int32_t Cout, int32_t C, int32_t ks) {
    if (Cout <= 0 || C <= 0 || (ks != 1 && ks != 3 && ks != 7)) return DCVC_E_ARG;
    const int64_t nco = (Cout + 31) / 32;
    return nco * ((C + 31) / 32) * ks * ks * 1024 + nco * 32;
}

extern "C" int dcvc_conv_wgrad(const dcvc_conv_wgrad_args *a, void *stream) {
    if (!a || !a->x || !a->dpre || !a->dw || !a->scratch || a->N <= 0 || a->C <= 0 || a->Cout <= 0) return DCVC_E_ARG;
    if ((a->ks != 1 && a->ks != 3 && a->ks != 7) || (a->stride != 1 && a->stride != 2) || (a->ks == 7 && a->stride != 1))
        return DCVC_E_ARG;
    if (a->precision != DCVC_PREC_FP32 && a->precision != DCVC_PREC_FP16X3) return DCVC_E_ARG;
    if (a->zs < 1 || (a->Ho - 1) * a->zs >= a->Hd || (a->Wo - 1) * a->zs >= a->Wd) return DCVC_E_ARG;
    if (a->cin_offset < 0 || a->cin_offset + a->C > a->Cin_total) return DCVC_E_ARG;
    const int pad = a->ks / 2;
    if ((a->Hin + 2 * pad - a->ks) / a->stride + 1 != a->Ho || (a->Win + 2 * pad - a->ks) / a->stride + 1 != a->Wo)
        return DCVC_E_ARG;
    const int64_t per_split = dcvc_conv_wgrad_scratch_min(a->Cout, a->C, a->ks);
    if (a->scratch_floats < per_split) return DCVC_E_ARG;
    WgradK k;
    k.x = a->x;
    k.x_cs = a->x_cs;
    k.C = a->C;
    k.in_act = a->in_act;
    k.in_slope = a->in_slope;
    k.dpre = a->dpre;
    k.dpre_cs = a->dpre_cs;
    k.zs = a->zs;
    k.Hd = a->Hd;
    k.Wd = a->Wd;
    k.N = a->N;
    k.Hin = a->Hin;
    k.Win = a->Win;
    k.Ho = a->Ho;
    k.Wo = a->Wo;
    k.Cout = a->Cout;
    k.scratch = a->scratch;
    k.bscratch = nullptr;
    k.nci = (a->C + 31) / 32;
    const int nco = (a->Cout + 31) / 32, nct = nco * k.nci;
    const int TW = a->stride == 1 ? 32 : 16;
    k.ntx = (a->Wo + TW - 1) / TW;
    k.nty = (a->Ho + 3) / 4;
    k.T = a->ks * a->ks;
    const int groups = a->ks == 7 ? 7 : 1;
    const int64_t ntiles = (int64_t)a->N * k.ntx * k.nty;
    // pixel splits: enough workgroups to fill the chip (~512), more only while every workgroup
    // still keeps >= 8 tiles to amortise its cross-wave reduction and its partial in scratch
    int64_t splits = (512 + (int64_t)nct * groups - 1) / ((int64_t)nct * groups);
    if (ntiles / 8 > splits) splits = ntiles / 8;
    if (splits > 256) splits = 256;
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    if (splits > a->scratch_floats / per_split) splits = a->scratch_floats / per_split;
    if (a->db) k.bscratch = a->scratch + (size_t)splits * nct * k.T * 1024;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)splits, (unsigned)nct, (unsigned)groups);
    // fast mode: bf16 hi/lo operands.  wgrad_bf16_kernel indexes dpre at (oy, ox): only without zero insertion (zs == 1)
    const bool split = a->precision == DCVC_PREC_FP16X3 && a->stride == 1 && a->zs == 1;
    if (split && a->ks == 3) hipLaunchKernelGGL((wgrad_bf16_kernel<3>), grid, dim3(256), 0, st, k);
    else if (split && a->ks == 7) hipLaunchKernelGGL((wgrad_bf16_kernel<7>), grid, dim3(256), 0, st, k);
    else if (split) hipLaunchKernelGGL((wgrad_bf16_kernel<1>), grid, dim3(256), 0, st, k);
    else if (a->ks == 3 && a->stride == 1) hipLaunchKernelGGL((wgrad_kernel<3, 1>), grid, dim3(256), 0, st, k);
    else if (a->ks == 3) hipLaunchKernelGGL((wgrad_kernel<3, 2>), grid, dim3(256), 0, st, k);
    else if (a->ks == 1 && a->stride == 1) hipLaunchKernelGGL((wgrad_kernel<1, 1>), grid, dim3(256), 0, st, k);
    else if (a->ks == 1) hipLaunchKernelGGL((wgrad_kernel<1, 2>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((wgrad_kernel<7, 1>), grid, dim3(256), 0, st, k);
    if (hipGetLastError() != hipSuccess) return DCVC_E_LAUNCH;
    const int64_t nel = (int64_t)nct * k.T * 1024;
    const int nb_main = (int)nblk(nel, 64), nb_bias = a->db ? (nco * 32 + 63) / 64 : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb_main + nb_bias), dim3(256), 0, st, a->scratch, a->dw, (int)splits, nct,
                       k.nci, k.T, a->Cout, a->C, a->Cin_total, a->cin_offset, a->overwrite, k.bscratch, a->db, nb_main);
    RET_LAUNCH();
}

extern "C" int dcvc_channel_dot(const float *a, int32_t a_cs, const float *b, int32_t b_cs, float *out, float *scratch,
                                int32_t N, int32_t HW, int32_t C, int32_t over_batch, int32_t accumulate, void *stream) {
    if (!a || !out || !scratch || N <= 0 || HW <= 0 || C <= 0) return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int Cl = C < 256 ? C : 256;
    const int G = 256 / Cl;
    int NB = (HW + G * 16 - 1) / (G * 16);
    NB = NB < 1 ? 1 : (NB > 256 ? 256 : NB);
    hipLaunchKernelGGL(channel_dot_partial, dim3(NB, N, (C + 255) / 256), dim3(256), 0, st, a, a_cs, b, b_cs, scratch, HW,
                       C, NB);
    hipLaunchKernelGGL(channel_dot_finish, dim3((C + 15) / 16, over_batch ? 1 : N), dim3(256), 0, st, scratch, out, N, C,
                       NB, over_batch, accumulate);
    RET_LAUNCH();
}

extern "C" int dcvc_mask_accumulate(const float *src, int32_t src_cs, const float *x, int32_t x_cs, float slope,
                                    float *dst, int32_t dst_cs, int64_t npix, int32_t C, void *stream) {
    if (!src || !dst || npix <= 0 || C <= 0) return DCVC_E_ARG;
    hipLaunchKernelGGL(mask_accumulate_kernel, dim3(nblk(npix * C, 256)), dim3(256), 0, (hipStream_t)stream, src, src_cs,
                       x, x_cs, slope, dst, dst_cs, npix, C);
    RET_LAUNCH();
}

extern "C" int dcvc_add_planes(const float *a, int32_t a_cs, const float *b, int32_t b_cs, float *out, int32_t out_cs,
                               int64_t npix, int32_t C, void *stream) {
    if (!a || !b || !out || npix <= 0 || C <= 0) return DCVC_E_ARG;
    hipLaunchKernelGGL(add_planes_kernel, dim3(nblk(npix * C, 256)), dim3(256), 0, (hipStream_t)stream, a, a_cs, b, b_cs,
                       out, out_cs, npix, C);
    RET_LAUNCH();
}

extern "C" int dcvc_warp_bwd(const float *src, int32_t src_cs, const float *flow, int32_t flow_cs, const float *dout,
                             int32_t dout_cs, float *dsrc, int32_t dsrc_cs, float *dflow, int32_t dflow_cs, int32_t N,
                             int32_t H, int32_t W, int32_t C, void *fix_scratch, void *stream) {
    if (!src || !flow || !dout || (!dsrc && !dflow) || N <= 0 || H <= 1 || W <= 1 || C <= 0) return DCVC_E_ARG;
    if (dsrc && !fix_scratch) return DCVC_E_ARG;
    int Cl = 1;
    while (Cl * 2 <= C && Cl < 64) Cl *= 2;  // threads per pixel (power of two <= 64)
    const int64_t npix = (int64_t)N * H * W;
    dim3 block(Cl, 256 / Cl);
    unsigned long long *fix = (unsigned long long *)fix_scratch;
    if (dsrc) {
        const int64_t nb = nblk(npix * C, 256 * 8);
        hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, (hipStream_t)stream, dout,
                           dout_cs, npix, C, (unsigned *)(fix + npix * C));
    }
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(nblk(npix, 256 / Cl)), block, 0, (hipStream_t)stream, src, src_cs, flow,
                       flow_cs, dout, dout_cs, dsrc, dsrc_cs, dflow, dflow_cs, N, H, W, C, fix);
    if (dsrc) {
        hipLaunchKernelGGL(fix_finish_kernel, dim3(nblk(npix * C, 256)), dim3(256), 0, (hipStream_t)stream, fix, dsrc, dsrc_cs,
                           npix, C);
        hipLaunchKernelGGL(fix_reset_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, fix + npix * C);
    }
    RET_LAUNCH();
}

extern "C" int dcvc_up2_bwd(const float *dout, int32_t dout_cs, float *dsrc, int32_t dsrc_cs, int32_t N, int32_t H,
                            int32_t W, int32_t C, float scale, void *stream) {
    if (!dout || !dsrc || N <= 0 || H <= 0 || W <= 0 || C <= 0) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(up2_bwd_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dout_cs, dsrc,
                       dsrc_cs, N, H, W, C, scale);
    RET_LAUNCH();
}

extern "C" int dcvc_down2_bwd(const float *dout, int32_t dout_cs, float *dsrc, int32_t dsrc_cs, int32_t N, int32_t H,
                              int32_t W, int32_t C, float scale, void *stream) {
    if (!dout || !dsrc || N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(down2_bwd_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dout_cs, dsrc,
                       dsrc_cs, N, H, W, C, scale * 0.25f);
    RET_LAUNCH();
}

extern "C" int dcvc_maxpool2_bwd(const float *src, int32_t src_cs, const float *dout, int32_t dout_cs, float *dsrc,
                                 int32_t dsrc_cs, int32_t N, int32_t H, int32_t W, int32_t C, void *stream) {
    if (!src || !dout || !dsrc || N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, src, src_cs, dout,
                       dout_cs, dsrc, dsrc_cs, N, H, W, C);
    RET_LAUNCH();
}

extern "C" int dcvc_se_bwd(const float *mean, const float *w1, const float *w2, const float *gate, const float *dgate,
                           float *dmean, float *dw1, float *dw2, int32_t N, int32_t C, int32_t Cr, void *stream) {
    if (!mean || !w1 || !w2 || !gate || !dgate || !dmean || C > 256 || Cr > 64 || Cr <= 0 || N <= 0) return DCVC_E_ARG;
    hipLaunchKernelGGL(se_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean, w1, w2, gate, dgate, dmean, dw1,
                       dw2, N, C, Cr);
    RET_LAUNCH();
}

extern "C" int dcvc_add_channel_vec(float *dst, int32_t dst_cs, const float *vec, float scale, int32_t N, int32_t HW,
                                    int32_t C, void *stream) {
    if (!dst || !vec || N <= 0 || HW <= 0 || C <= 0) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * HW * C;
    hipLaunchKernelGGL(add_channel_vec_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, dst, dst_cs,
                       vec, scale, (int64_t)HW, C, total);
    RET_LAUNCH();
}

extern "C" int dcvc_scale_channels_bwd(const float *dout, int32_t dout_cs, float *dsrc, int32_t dsrc_cs,
                                       const float *q_basic, const float *q_scale, int32_t mode, int32_t N, int32_t HW,
                                       int32_t C, void *stream) {
    if (!dout || !dsrc || !q_basic || !q_scale) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * HW * C;
    hipLaunchKernelGGL(scale_channels_bwd_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, dout,
                       dout_cs, dsrc, dsrc_cs, q_basic, q_scale, mode, (int64_t)HW, C, total);
    RET_LAUNCH();
}

extern "C" int dcvc_q_finish(const float *dq_mul, const float *s_div, const float *q_basic, const float *q_scale,
                             float *dq_basic, float *dq_scale, int32_t N, int32_t C, void *stream) {
    if ((!dq_mul && !s_div) || !q_basic || !q_scale || N <= 0 || C <= 0) return DCVC_E_ARG;
    hipLaunchKernelGGL(q_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dq_mul, s_div, q_basic, q_scale,
                       dq_basic, dq_scale, N, C);
    RET_LAUNCH();
}

extern "C" int dcvc_dual_prior_bwd(const dcvc_dual_prior_bwd_args *a, void *stream) {
    if (!a || (a->C & 1) || a->N <= 0 || (a->step != 0 && a->step != 1)) return DCVC_E_ARG;
    if (a->step == 1 && !a->dspatial) return DCVC_E_ARG;
    if (a->step == 0 && (!a->y || !a->fusion || !a->y_hat || !a->dfusion || !a->q_basic || !a->q_scale)) return DCVC_E_ARG;
    const int64_t total = (int64_t)a->N * a->H * a->W * a->C;
    hipLaunchKernelGGL(dual_prior_bwd_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, *a, total);
    RET_LAUNCH();
}

extern "C" int dcvc_scale_bits_bwd(const float *y, const float *scales_hat, const float *g, float *dy, float *dscales,
                                   int32_t kind, int32_t N, int64_t per_sample, void *stream) {
    if (!y || !scales_hat || !g || !dy || !dscales || N <= 0 || per_sample <= 0 || (kind != 0 && kind != 1)) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * per_sample;
    hipLaunchKernelGGL(scale_bits_bwd_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, y, scales_hat, g,
                       dy, dscales, kind, per_sample, total);
    RET_LAUNCH();
}

extern "C" int dcvc_factorized_bits_bwd(const float *z, int32_t z_cs, const float *params, const float *g, float *dz,
                                        int32_t dz_cs, float *dparams, int32_t N, int32_t HW, int32_t C, void *stream) {
    if (!z || !params || !g || (!dz && !dparams) || N <= 0 || HW <= 0 || C <= 0) return DCVC_E_ARG;
    hipLaunchKernelGGL(factorized_bits_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, z, z_cs, params, g, dz,
                       dz_cs, dparams, N, (int64_t)HW, C);
    RET_LAUNCH();
}

extern "C" int dcvc_sq_err_bwd(const float *a, int32_t a_cs, const float *b, int32_t b_cs, const float *g, float *da,
                               int32_t da_cs, int32_t N, int32_t HW, int32_t C, void *stream) {
    if (!a || !b || !g || !da || N <= 0 || HW <= 0 || C <= 0) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * HW * C;
    hipLaunchKernelGGL(sq_err_bwd_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, a, a_cs, b, b_cs, g,
                       da, da_cs, (int64_t)HW, C, total);
    RET_LAUNCH();
}
