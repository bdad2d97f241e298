// conv_s16.hip -- persistent 3x3 stride-1 convolution over activations stored PRE-SPLIT in HBM.
//
// Same arithmetic as conv_mfma.hip's DCVC_PREC_FP16X3 mode (x*w ~= xh*wh + xh*wl + xl*wh on
// v_mfma_f32_32x32x16_f16, fp32 accumulation, identical product order), for the layers that
// dominate a P picture: /root/reference/DCVC_HEM/src/models/video_net.py:74-96 (ResBlock),
// src/layers/layers.py:42-127 (ResidualBlock*), video_model.py:17-128 (feature extractor,
// context fusion, contextual encoder / decoder, recon generation).
//
// Why a second kernel.  conv_mfma reads fp32 activations, so every consumer converts every patch
// element to (hi, lo) in VALU and stages it through prefetch registers; profiling (DESIGN.md
// section 4) shows the MFMA pipe idle more than half the time behind that staging and behind
// an epilogue no other work overlaps.  Here
//   * the PRODUCER's epilogue writes the activation already split ("S16" layout below, same
//     4 bytes per element), optionally with the consumer's input activation applied, so that
//   * patches and filter slabs go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging
//     registers, no converts, loads for step s+1 in flight while step s multiplies;
//   * a workgroup is PERSISTENT (one per CU, 8 waves, 150 KB of LDS as two patch + two filter
//     buffers) and walks 16x32-pixel tiles; the first patch of the next tile is requested before
//     the current tile's last multiply phase, and the epilogue's stores are fire-and-forget, so
//     HBM traffic of one tile overlaps the MFMAs of the next;
//   * the GEMM is turned around (M = output channels, N = 32 pixels of a row): the accumulator
//     then holds 16 consecutive channels of ONE pixel per lane -> every epilogue access is a
//     contiguous 64-byte run per lane (4 x dwordx4) without an LDS transpose;
//   * tiles are dealt to workgroups in per-XCD bands, so halo rows are re-read from that XCD's L2.
//
// S16 layout of a C-channel tensor (C % 16 == 0): PLANAR.  Every 16-channel chunk p of image n is
// four (H, W) planes of 16-byte entries -- slot 0: fp16 hi of channels 0-7 of the chunk, slot 1: hi
// of channels 8-15, slot 2: lo of 0-7, slot 3: lo of 8-15 (hi = fp16(8 v), lo = fp16(8 v - hi),
// ACT_SCALE = 8 as in conv_mfma.hip) -- i.e. exactly the 16-byte MFMA operand fragments:
//     base + (((n * cs/16 + p) * 4 + slot) * H*W + y*W + x) * 16
// where cs is the channel count of the underlying buffer (a channel slice = a run of planes).  A
// lane of the accumulator owns one pixel, so a wave's store of one slot is 32 x 16 B = 512 B
// contiguous per lane half, and a patch row of a slot is one contiguous run for the DMA.
//
// LDS images.  Patch: slot-planar like the tensor, [4 slots][(16+2) x (32+2) pixels][16 B], linear
// in DMA order (a wave-instruction copies runs of a 34-pixel patch row of one slot plane: whole
// cache lines); consecutive lanes read consecutive 16 B, so every ds_read_b128 is conflict-free
// without a swizzle.  Filter slab: [tap][hi h0, hi h1, lo h0, lo h1]
// [64 output rows][8 fp16], copied linearly from the host-packed weights.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "dcvc_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr float ACT_SCALE = 8.f, WGT_SCALE = 64.f, F16_MAX = 65504.f;
constexpr int KC = 16;

__device__ uint4 g_zero_page[16];  // 256 B of zeros: source of out-of-picture patch pixels

struct S16K {
    const char *seg_ptr[DCVC_MAX_SEG];
    int seg_chunks[DCVC_MAX_SEG];  // 16-channel chunks per segment
    int seg_planes[DCVC_MAX_SEG];  // planes per image of the segment's underlying buffer (cs / 16)
    int nseg, nchunks;
    int N, H, W;
    const char *wpack;
    const float *bpack;
    int Cout, nblk;
    float *out;
    int out_cs, out_act;
    float out_slope;
    char *out16;
    int out16_cs, out16_act;
    float out16_slope;
    int ps;
    const char *res;
    int res_cs, res_fmt;
    const float *res_gate;
    const char *res2;
    int res2_cs, res2_fmt;
    int *status;
    int ntx, nty;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

__device__ __forceinline__ void keep_alive(const f32x16 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#endif
}

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// 16 consecutive channels of one pixel TIMES ACT_SCALE, from an fp32 NHWC or an S16 source.  pixel: index over
// (n, y, x); img / pin: its image and in-image parts; hw: pixels per image.  (hi + lo is exact in fp32.)
__device__ __forceinline__ void load16x8(const char *base, size_t pixel, int img, size_t pin, size_t hw, int cs, int fmt,
                                         int cf, float *v) {
    if (fmt == DCVC_FMT_F32) {
        const f32x4 *p = (const f32x4 *)(base + (pixel * cs + cf) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 t = p[i] * ACT_SCALE;
            v[4 * i] = t[0], v[4 * i + 1] = t[1], v[4 * i + 2] = t[2], v[4 * i + 3] = t[3];
        }
    } else {
        const char *p = base + ((((size_t)img * (cs >> 4) + (cf >> 4)) * 4) * hw + pin) * 16;
        const f16x8 h0 = *(const f16x8 *)p, h1 = *(const f16x8 *)(p + hw * 16), l0 = *(const f16x8 *)(p + hw * 32),
                    l1 = *(const f16x8 *)(p + hw * 48);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = (float)h0[i] + (float)l0[i];
            v[8 + i] = (float)h1[i] + (float)l1[i];
        }
    }
}

// PROBE is 0 in the product library; tools/probes/conv_s16_probe.hip instantiates the ablations
// (1: no DMA after the first step, 2: no MFMA phase, 4: no epilogue) to see what each phase costs.
template <int KS, int NT, int PROBE = 0>
__global__ __launch_bounds__(512, 2) void conv_s16_kernel(const S16K a) {
    constexpr int BH = 16, BW = 32, PAD = KS / 2, PH = BH + KS - 1, PW = BW + KS - 1, T = KS * KS, CB = 32 * NT;
    constexpr int NPIX = PH * PW, PSLOTS = NPIX * 4, NWI_P = (PSLOTS + 63) / 64, PATCH_BYTES = NWI_P * 1024;
    constexpr int FILT_BYTES = T * 4 * CB * 16, NWI_F = FILT_BYTES / 1024;
    constexpr int RP = (NWI_P + 7) / 8, RF = (NWI_F + 7) / 8;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char *const patch = lds;                    // 2 x PATCH_BYTES
    char *const filt = lds + 2 * PATCH_BYTES;   // 2 x FILT_BYTES

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hh = lane >> 5;
    const int r0 = 2 * wave;

    // ---- this workgroup's tiles: per-XCD bands (blocks g and g+8 share an XCD's L2) when the grid allows
    const int per_img = a.ntx * a.nty, total = a.nblk * a.N * per_img;
    int first, step, count;
    {
        const int G = gridDim.x, g = blockIdx.x;
        if ((G & 7) == 0) {
            const int gx = G >> 3, x = g & 7, j = g >> 3, t8 = (total + 7) >> 3;
            const int lo = x * t8, hi = min(lo + t8, total);
            first = lo + j, step = gx;
            count = first < hi ? (hi - first + gx - 1) / gx : 0;
        } else {
            first = g, step = G;
            count = first < total ? (total - first + G - 1) / G : 0;
        }
    }
    if (count == 0) return;

    // ---- DMA lane geometry: entry gs = wi*64 + lane of the slot-planar patch image [slot][pixel][16 B]
    struct Tile {
        int blk, img, y0, x0;
    };
    auto decode = [&](int w) {
        Tile t;
        t.blk = w / (a.N * per_img);
        int r = w - t.blk * (a.N * per_img);
        t.img = r / per_img;
        r -= t.img * per_img;
        const int ty = r / a.ntx;
        t.y0 = ty * BH;
        t.x0 = (r - ty * a.ntx) * BW;
        return t;
    };
    const unsigned hw_in = (unsigned)(a.H * a.W);
    int pixoff[RP];  // (slot * H*W + pixel) of the source entry, or -1: out of the picture -> zero page
    auto setup_dma = [&](const Tile &t) {
#pragma unroll
        for (int u = 0; u < RP; ++u) {
            const int gs = (wave + 8 * u) * 64 + lane;
            const int slot = gs / NPIX, q = gs - slot * NPIX;
            const int py = q / PW, px = q - py * PW;
            const int gy = t.y0 - PAD + py, gx = t.x0 - PAD + px;
            const bool ok = gs < PSLOTS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            pixoff[u] = ok ? (int)(slot * hw_in) + gy * a.W + gx : -1;
        }
    };
    // DMA of one step, issued round by round from inside the multiply phase (an LDS-DMA costs its wave
    // 60-180 issue cycles: spread between the MFMAs, and at different points for the two waves of a SIMD,
    // the partner's MFMAs cover them).  chunk cg of the concatenated input -> (segment, chunk inside it)
    const char *dma_sp = nullptr, *dma_fp = nullptr;
    char *dma_pd = nullptr, *dma_fd = nullptr;
    auto dma_begin = [&](const Tile &t, int cg, int buf) {
        int s = 0, c = cg;
        while (c >= a.seg_chunks[s]) c -= a.seg_chunks[s], ++s;
        dma_sp = a.seg_ptr[s] + (((size_t)t.img * a.seg_planes[s] + c) * 4) * ((size_t)hw_in * 16);
        dma_fp = a.wpack + ((size_t)(t.blk * a.nchunks + cg)) * FILT_BYTES + lane * 16;
        dma_pd = patch + buf * PATCH_BYTES;
        dma_fd = filt + buf * FILT_BYTES;
    };
    auto dma_round = [&](int u) {  // u: compile-time round index
        const int wi = wave + 8 * u;
        if (u < RP && wi < NWI_P) {
            const char *src = pixoff[u] >= 0 ? dma_sp + (size_t)((unsigned)pixoff[u] * 16u)
                                             : (const char *)g_zero_page + (lane & 15) * 16;
            glds16(src, dma_pd + wi * 1024);
        }
        if (u < RF && wi < NWI_F) glds16(dma_fp + wi * 1024, dma_fd + wi * 1024);
    };
    constexpr int NR = RP > RF ? RP : RF;
    static_assert(NR <= T, "one DMA round per tap");

    // per-lane LDS read offsets: the hi fragment of a pixel is entry (slot hh, pixel), lo two slot planes on
    const int plane_off = (hh * NPIX + r0 * PW + col) * 16;
    const int wlane = (hh * CB + col) * 16;

    Tile cur = decode(first);
    setup_dma(cur);
    dma_begin(cur, 0, 0);
#pragma unroll
    for (int u = 0; u < NR; ++u) dma_round(u);
    int sbuf = 0;  // buffer (patch and filter alike) holding the step about to be multiplied
    const int late = wave >> 2;  // waves w and w+4 share a SIMD: they issue their DMA rounds half a tap apart

    for (int it = 0; it < count; ++it) {
        f32x16 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        Tile nxt = cur;
        for (int cg = 0; cg < a.nchunks; ++cg) {
            // this wave's share of step (it, cg) has landed; after the barrier everybody's has, and
            // every wave is done reading the other buffer (its previous step)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            bool more = false;
            if (cg + 1 < a.nchunks) {
                dma_begin(cur, cg + 1, sbuf ^ 1);
                more = true;
            } else if (it + 1 < count) {
                nxt = decode(first + (it + 1) * step);
                setup_dma(nxt);
                dma_begin(nxt, 0, sbuf ^ 1);
                more = true;
            }
            if (PROBE & 1) more = false;
            const char *pb = patch + sbuf * PATCH_BYTES + plane_off;
            const char *fb = filt + sbuf * FILT_BYTES + wlane;
#pragma unroll
            for (int t = 0; t < ((PROBE & 2) ? 0 : T); ++t) {
                const int ky = t / KS, kx = t % KS;
                f16x8 xh[2], xl[2], wh[NT], wl[NT];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    xh[m] = *(const f16x8 *)(pb + ((m + ky) * PW + kx) * 16);
                    xl[m] = *(const f16x8 *)(pb + ((m + ky) * PW + kx + 2 * NPIX) * 16);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    wh[n] = *(const f16x8 *)(fb + ((t * 4) * CB + n * 32) * 16);
                    wl[n] = *(const f16x8 *)(fb + ((t * 4 + 2) * CB + n * 32) * 16);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[n], xh[m], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[n], xh[m], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[n], xl[m], acc[m][n], 0, 0, 0);
                    }
                    // the step's DMA rounds ride in the first taps, so that the last one still has most of
                    // the multiply phase to land
                    if (t < NR && more && m == late) dma_round(t);
                }
            }
            if ((PROBE & 2) && more) {
#pragma unroll
                for (int u = 0; u < NR; ++u) dma_round(u);
            }
            sbuf ^= 1;
        }

        // ---- epilogue: each lane owns 16 consecutive (packed-order) channels of one pixel per (m, n).
        // Everything is computed TIMES ACT_SCALE (a power of two: bit-identical to scaling at the end), which
        // is the domain the s16 operands live in.
        const int Cq = a.Cout >> 2;
        const int Ho = a.ps ? a.H * 2 : a.H, Wo = a.ps ? a.W * 2 : a.W;
        const int Cfin = a.ps ? Cq : a.Cout;
        const size_t hw = (size_t)Ho * Wo;
        const float inv_scale = 1.f / WGT_SCALE;  // accumulators carry ACT_SCALE * WGT_SCALE
        bool sat = false;
        if (PROBE & 4) {  // keep the accumulators alive without touching memory
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) keep_alive(acc[m][n]);
        }
#pragma unroll
        for (int m = 0; m < ((PROBE & 4) ? 0 : 2); ++m) {
            const int oy = cur.y0 + r0 + m, ox = cur.x0 + col;
            // the residual loads of a row go out before its first store (res may alias an output: each element
            // is read and written by the same lane), so their latency is paid once per row, not per channel tile
            float rres[NT][16];
            if (a.res) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int cb = cur.blk * CB + n * 32 + 16 * hh;
                    if (!(oy < a.H && ox < a.W && cb < a.Cout)) continue;
                    int cf = cb, dy = 0, dx = 0;
                    if (a.ps) {
                        const int sub = cb / Cq;
                        cf = cb - sub * Cq;
                        dy = sub >> 1, dx = sub & 1;
                    }
                    const size_t pin = a.ps ? ((size_t)(2 * oy + dy) * Wo + 2 * ox + dx) : ((size_t)oy * Wo + ox);
                    load16x8(a.res, (size_t)cur.img * hw + pin, cur.img, pin, hw, a.res_cs, a.res_fmt, cf, rres[n]);
                }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int cb = cur.blk * CB + n * 32 + 16 * hh;
                if (!(oy < a.H && ox < a.W && cb < a.Cout)) continue;
                int cf = cb, dy = 0, dx = 0;
                if (a.ps) {
                    const int sub = cb / Cq;
                    cf = cb - sub * Cq;
                    dy = sub >> 1, dx = sub & 1;
                }
                const size_t pin = a.ps ? ((size_t)(2 * oy + dy) * Wo + 2 * ox + dx) : ((size_t)oy * Wo + ox);
                const size_t pixel = (size_t)cur.img * hw + pin;
                float v[16], rv2[16];
                const float *rv = rres[n];
                if (a.res2) load16x8(a.res2, pixel, cur.img, pin, hw, a.res2_cs, a.res2_fmt, cf, rv2);
                {
                    const f32x4 *bp = (const f32x4 *)(a.bpack + cb);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 b4 = bp[i] * ACT_SCALE;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * i + e] = acc[m][n][4 * i + e] * inv_scale + b4[e];
                    }
                }
                if (a.out_act == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = lrelu(v[i], a.out_slope);
                } else if (a.out_act == 2) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = fminf(fmaxf(v[i], 0.f), ACT_SCALE);
                }
                if (a.res) {
                    if (a.res_gate) {
                        const float *gp = a.res_gate + (size_t)cur.img * Cfin + cf;
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(rv[i], gp[i], v[i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] += rv[i];
                    }
                }
                if (a.res2) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = rv2[i] + v[i];
                }
                if (a.out) {
                    f32x4 *op = (f32x4 *)(a.out + pixel * a.out_cs + cf);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        op[i] = (f32x4){v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]} * (1.f / ACT_SCALE);
                }
                if (a.out16) {
                    f16x8 h[2], l[2];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float s = a.out16_act ? lrelu(v[i], a.out16_slope) : v[i];
                        sat |= !(fabsf(s) <= F16_MAX);
                        s = __builtin_amdgcn_fmed3f(s, -F16_MAX, F16_MAX);
                        const _Float16 hi = (_Float16)s;
                        h[i >> 3][i & 7] = hi;
                        l[i >> 3][i & 7] = (_Float16)(s - (float)hi);
                    }
                    char *op = a.out16 + ((((size_t)cur.img * (a.out16_cs >> 4) + (cf >> 4)) * 4) * hw + pin) * 16;
                    *(f16x8 *)op = h[0];
                    *(f16x8 *)(op + hw * 16) = h[1];
                    *(f16x8 *)(op + hw * 32) = l[0];
                    *(f16x8 *)(op + hw * 48) = l[1];
                }
            }
        }
        if (sat && a.status) atomicOr(a.status, DCVC_STATUS_ACT_SATURATED);
        cur = nxt;
    }
}

int g_cus = 0;

template <int NT, int PROBE = 0>
int launch_s16(const S16K &k, hipStream_t st) {
    constexpr int KS = 3, PH = 16 + KS - 1, PW = 32 + KS - 1, CB = 32 * NT;
    constexpr int PATCH_BYTES = ((PH * PW * 4 + 63) / 64) * 1024, FILT_BYTES = KS * KS * 4 * CB * 16;
    constexpr int LDS = 2 * (PATCH_BYTES + FILT_BYTES);
    static int attr = -1;
    if (attr < 0)
        attr = hipFuncSetAttribute((const void *)conv_s16_kernel<KS, NT, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) ==
                       hipSuccess
                   ? 1
                   : 0;
    if (!attr) return DCVC_E_LAUNCH;
    if (!g_cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return DCVC_E_LAUNCH;
        g_cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const int total = k.nblk * k.N * k.ntx * k.nty;
    int G = total < g_cus ? total : g_cus;
    if (G >= 8) G &= ~7;
    hipLaunchKernelGGL((conv_s16_kernel<KS, NT, PROBE>), dim3(G), dim3(512), LDS, st, k);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline bool al64(const void *p) { return (((uintptr_t)p) & 63) == 0; }

// ---- fp32 <-> S16 conversion of a strided NHWC tensor (boundary of the format) -----------------
// one thread per (chunk, pixel), pixels fastest: the s16 side is contiguous, the fp32 side strided
__global__ void s16_pack_kernel(const float *__restrict__ src, int src_cs, char *__restrict__ out, int out_cs, int64_t npix,
                                int64_t hw, int C, int act, float slope, int *status) {
    const int nch = (C + 15) >> 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * nch) return;
    const int ch = (int)(gid / npix);
    const int64_t pix = gid - (int64_t)ch * npix;
    const int64_t img = pix / hw, pin = pix - img * hw;
    f16x8 h[2], l[2];
    bool sat = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = ch * 16 + i;
        float s = c < C ? src[pix * src_cs + c] : 0.f;
        if (act) s = lrelu(s, slope);
        s *= ACT_SCALE;
        sat |= !(fabsf(s) <= F16_MAX);
        s = __builtin_amdgcn_fmed3f(s, -F16_MAX, F16_MAX);
        const _Float16 hi = (_Float16)s;
        h[i >> 3][i & 7] = hi;
        l[i >> 3][i & 7] = (_Float16)(s - (float)hi);
    }
    char *op = out + (((img * (out_cs >> 4) + ch) * 4) * hw + pin) * 16;
    *(f16x8 *)op = h[0];
    *(f16x8 *)(op + hw * 16) = h[1];
    *(f16x8 *)(op + hw * 32) = l[0];
    *(f16x8 *)(op + hw * 48) = l[1];
    if (sat && status) atomicOr(status, DCVC_STATUS_ACT_SATURATED);
}

__global__ void s16_unpack_kernel(const char *__restrict__ src, int src_cs, float *__restrict__ out, int out_cs, int64_t npix,
                                  int64_t hw, int C) {
    const int nch = (C + 15) >> 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * nch) return;
    const int ch = (int)(gid / npix);
    const int64_t pix = gid - (int64_t)ch * npix;
    const int64_t img = pix / hw, pin = pix - img * hw;
    float v[16];
    load16x8(src, (size_t)pix, (int)img, (size_t)pin, (size_t)hw, src_cs, DCVC_FMT_S16, ch * 16, v);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (ch * 16 + i < C) out[pix * out_cs + ch * 16 + i] = v[i] * (1.f / ACT_SCALE);
}

}  // namespace

extern "C" int64_t dcvc_conv_s16_pack_bytes(int32_t Cout, int32_t ks, int32_t nseg, const int32_t *seg_C, int32_t *cout_pad) {
    if (Cout <= 0 || nseg <= 0 || nseg > DCVC_MAX_SEG || ks != 3 || (Cout & 15)) return DCVC_E_ARG;
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) {
        if (seg_C[s] <= 0 || (seg_C[s] & 15)) return DCVC_E_ARG;
        chunks += seg_C[s] / KC;
    }
    const int cp = round_up(Cout, 32);
    if (cout_pad) *cout_pad = cp;
    return (int64_t)chunks * ks * ks * 4 * cp * 16;
}

// wpack as fp16: [block b][chunk][tap][hi h0, hi h1, lo h0, lo h1][CB rows][8], CB = 64 when Cout_pad % 64 == 0
// else 32.  Row rho of a 32-row MFMA tile carries the packed-order channel 16*((rho>>2)&1) + (rho&3) + 4*(rho>>3)
// of that tile, so that lane half h of the accumulator holds channels 16h .. 16h+15 in its 16 registers.
extern "C" int dcvc_conv_s16_pack_weights(const float *w, const float *b, int32_t Cout, int32_t ks, int32_t nseg,
                                          const int32_t *seg_C, int32_t pixel_shuffle, void *wpack, float *bpack) {
    int32_t cp = 0;
    const int64_t total = dcvc_conv_s16_pack_bytes(Cout, ks, nseg, seg_C, &cp);
    if (total < 0 || !w || !wpack || !bpack) return DCVC_E_ARG;
    if (pixel_shuffle && ((Cout & 3) || ((Cout / 4) & 15))) return DCVC_E_ARG;
    const int T = ks * ks, CB = (cp % 64 == 0) ? 64 : 32, nblk = cp / CB;
    int Cin = 0, nchunks = 0;
    for (int s = 0; s < nseg; ++s) Cin += seg_C[s], nchunks += seg_C[s] / KC;
    memset(wpack, 0, (size_t)total);
    memset(bpack, 0, (size_t)cp * sizeof(float));
    const int Cq = Cout / 4;
    _Float16 *base = (_Float16 *)wpack;
    int status = DCVC_OK;
    for (int n = 0; n < Cout; ++n) {
        const int np = pixel_shuffle ? (n & 3) * Cq + (n >> 2) : n;  // packed-order position of channel n
        bpack[np] = b ? b[n] : 0.f;
        const int blk = np / CB, tile = (np % CB) / 32, ch = np % 32;
        // inverse of ch(rho): ch = 16*hbit + 4*g + e  ->  rho = e + 4*hbit + 8*g
        const int rho = (ch & 3) + 4 * (ch >> 4) + 8 * ((ch >> 2) & 3);
        const int row = tile * 32 + rho;
        for (int cg = 0; cg < nchunks; ++cg)
            for (int t = 0; t < T; ++t)
                for (int cc = 0; cc < KC; ++cc) {
                    float sv = w[((size_t)n * Cin + cg * KC + cc) * T + t] * WGT_SCALE;
                    if (!(sv <= F16_MAX && sv >= -F16_MAX)) {
                        status = DCVC_E_RANGE;
                        sv = sv > 0 ? F16_MAX : -F16_MAX;
                    }
                    const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
                    const int h = cc >> 3, jj = cc & 7;
                    const size_t slab = (((size_t)blk * nchunks + cg) * T + t) * 4;
                    base[((slab + h) * CB + row) * 8 + jj] = hi;
                    base[((slab + 2 + h) * CB + row) * 8 + jj] = lo;
                }
    }
    return status;
}

namespace {
int build_s16k(const dcvc_conv_s16_args *a, S16K &k, int &CB) {
    if (!a || a->nseg < 1 || a->nseg > DCVC_MAX_SEG || !a->wpack || !a->bpack || (!a->out && !a->out16)) return DCVC_E_ARG;
    if (a->ks != 3 || a->Cout <= 0 || (a->Cout & 15) || a->Cout_pad != round_up(a->Cout, 32)) return DCVC_E_ARG;
    if (a->N <= 0 || a->H <= 0 || a->W <= 0) return DCVC_E_ARG;
    const int cfin = a->pixel_shuffle ? a->Cout / 4 : a->Cout;
    if (a->pixel_shuffle && ((a->Cout & 3) || (cfin & 15))) return DCVC_E_ARG;
    memset(&k, 0, sizeof(k));
    for (int s = 0; s < a->nseg; ++s) {
        if (!a->seg[s].ptr || !al64(a->seg[s].ptr) || (a->seg[s].C & 15) || a->seg[s].C <= 0 || (a->seg[s].cs & 15) ||
            a->seg[s].cs < a->seg[s].C)
            return DCVC_E_ARG;
        if ((int64_t)a->H * a->W * 64 >= (1ll << 32)) return DCVC_E_ARG;  // 32-bit in-plane byte offsets
        k.seg_ptr[s] = (const char *)a->seg[s].ptr;
        k.seg_chunks[s] = a->seg[s].C / KC;
        k.seg_planes[s] = a->seg[s].cs / KC;
        k.nchunks += k.seg_chunks[s];
    }
    auto ok_f32 = [](const void *p, int cs) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (cs & 3) == 0); };
    auto ok_s16 = [](const void *p, int cs) { return p == nullptr || ((((uintptr_t)p) & 63) == 0 && (cs & 15) == 0); };
    if (!ok_f32(a->out, a->out_cs) || !ok_s16(a->out16, a->out16_cs)) return DCVC_E_ARG;
    if (!(a->res_fmt == DCVC_FMT_S16 ? ok_s16(a->res, a->res_cs) : ok_f32(a->res, a->res_cs))) return DCVC_E_ARG;
    if (!(a->res2_fmt == DCVC_FMT_S16 ? ok_s16(a->res2, a->res2_cs) : ok_f32(a->res2, a->res2_cs))) return DCVC_E_ARG;
    if (a->res_gate && !a->res) return DCVC_E_ARG;
    k.nseg = a->nseg;
    k.N = a->N, k.H = a->H, k.W = a->W;
    k.wpack = (const char *)a->wpack;
    k.bpack = a->bpack;
    k.Cout = a->Cout;
    CB = (a->Cout_pad % 64 == 0) ? 64 : 32;
    k.nblk = a->Cout_pad / CB;
    k.out = a->out, k.out_cs = a->out_cs, k.out_act = a->out_act, k.out_slope = a->out_slope;
    k.out16 = (char *)a->out16, k.out16_cs = a->out16_cs, k.out16_act = a->out16_act, k.out16_slope = a->out16_slope;
    k.ps = a->pixel_shuffle;
    k.res = (const char *)a->res, k.res_cs = a->res_cs, k.res_fmt = a->res_fmt;
    k.res_gate = a->res_gate;
    k.res2 = (const char *)a->res2, k.res2_cs = a->res2_cs, k.res2_fmt = a->res2_fmt;
    k.status = a->status;
    k.ntx = (a->W + 31) / 32, k.nty = (a->H + 15) / 16;
    return DCVC_OK;
}
}  // namespace

extern "C" int dcvc_conv2d_s16(const dcvc_conv_s16_args *a, void *stream) {
    S16K k;
    int CB = 0;
    const int rc = build_s16k(a, k, CB);
    if (rc != DCVC_OK) return rc;
    return CB == 64 ? launch_s16<2>(k, (hipStream_t)stream) : launch_s16<1>(k, (hipStream_t)stream);
}

extern "C" int dcvc_s16_pack(const float *src, int32_t src_cs, void *out, int32_t out_cs, int32_t N, int64_t hw, int32_t C,
                             int32_t act, float slope, int32_t *status, void *stream) {
    if (!src || !out || C <= 0 || N < 0 || hw < 0 || (out_cs & 15) || out_cs < round_up(C, 16) || !al64(out)) return DCVC_E_ARG;
    const int64_t npix = (int64_t)N * hw;
    if (npix == 0) return DCVC_OK;
    const int64_t total = npix * ((C + 15) / 16);
    hipLaunchKernelGGL(s16_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, src_cs,
                       (char *)out, out_cs, npix, hw, C, act, slope, status);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}

extern "C" int dcvc_s16_unpack(const void *src, int32_t src_cs, float *out, int32_t out_cs, int32_t N, int64_t hw, int32_t C,
                               void *stream) {
    if (!src || !out || C <= 0 || N < 0 || hw < 0 || (src_cs & 15) || src_cs < round_up(C, 16) || !al64(src)) return DCVC_E_ARG;
    const int64_t npix = (int64_t)N * hw;
    if (npix == 0) return DCVC_OK;
    const int64_t total = npix * ((C + 15) / 16);
    hipLaunchKernelGGL(s16_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const char *)src, src_cs, out, out_cs, npix, hw, C);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}
