// conv_mfma.hip -- implicit-GEMM 2-D convolution on the CDNA4 matrix cores, exact fp32.
//
// Replaces every nn.Conv2d of the reference's P-/I-frame networks together with the ops the
// reference runs around it as separate kernels: torch.cat of the inputs (multi-segment
// prologue), the (Leaky)ReLU before/after, the residual add, the SE gate and
// nn.PixelShuffle(2) (permuting epilogue).  /root/reference/DCVC_HEM/src/layers/layers.py:18-127,
// src/models/video_net.py:74-115,165-223, src/models/video_model.py:17-128.
//
// GEMM view:  Out[pixel][cout] = sum_{tap, cin} In[pixel + tap][cin] * W[tap][cin][cout]
//   M = 32 consecutive output pixels of one row  (MFMA rows)
//   N = 32 output channels                       (MFMA columns -> NHWC stores of 128 B)
//   K = 16 input channels x KSxKS taps per chunk, fed two at a time to
//       v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: bitwise an fmaf chain, so the
//       encoder and the decoder passes of the codec see identical numbers run to run).
// A 256-thread workgroup (4 waves) owns a (4*RPW rows) x 32 px x (32*NT channels) output
// tile; each wave keeps RPW x NT accumulators of 32x32.  Per 16-channel chunk the input
// patch (with halo) and the filter slab are staged in LDS:
//   patch [PH][PW][16 (+4 pad)] floats : row stride 20 dwords makes the per-lane
//                                        ds_read_b128 of 4 channels conflict-free for
//                                        stride-1 convs (bank start = 4*(5*lane mod 16))
//   wl    [taps][4][32*NT][4]   floats : packed on the host so that the copy is linear and
//                                        consecutive lanes read consecutive 16 B.
// K order inside a chunk is permuted (lane-half h of the MFMA takes channels 8*k2+4*h+j):
// A and B use the same permutation, so the sum is unchanged.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "dcvc_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 16;   // channels per K chunk
constexpr int LDK = 20;  // padded channel stride of the LDS patch, in floats

// Split-fp16 mode ("fp16x3"): every fp32 operand v is carried as hi = fp16(v * 2^s) and
// lo = fp16(v * 2^s - hi); x*w ~= xh*wh + xh*wl + xl*wh on v_mfma_f32_32x32x16_f16 with fp32
// accumulation (products of fp16 are exact in fp32; the dropped xl*wl is <= 2^-22 |x w|).
// gfx950's MFMA honours fp16 subnormals (tools/probes/mfma_f16_subnormal.hip), so lo needs no
// separate scale: representation error <= max(2^-22 |v|, 2^-25 / 2^s).  The power-of-two
// pre-scales keep typical activations / weights in fp16's normal range and are undone exactly
// in the epilogue.  3 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64 cycles per 16-deep K step.
constexpr float ACT_SCALE = 8.f;     // activations: |x| < 8188 representable
constexpr float WGT_SCALE = 64.f;    // weights:     |w| < 1023 representable
constexpr float F16_MAX = 65504.f;

struct ConvK {
    const float *seg_ptr[DCVC_MAX_SEG];
    int seg_C[DCVC_MAX_SEG];
    int seg_cs[DCVC_MAX_SEG];
    int nseg;
    int Hin, Win, Hout, Wout;
    int in_act;
    float in_slope;
    const float *wpack;
    const float *bpack;
    int Cout, Cout_pad;
    float *out;
    int out_cs;
    int out_act;
    float out_slope;
    int ps;
    const float *res;
    int res_cs;
    const float *res_gate;
    const float *res2;
    int res2_cs;
    int vec_epi;  // 1: every output / residual row is 16-byte addressable in groups of 4 channels
};

__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : v * slope; }

__device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    v = fminf(fmaxf(v * ACT_SCALE, -F16_MAX), F16_MAX);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

#ifdef PROBE_TPS3
#define CLASSIC_WAVES_PER_SIMD 3
#else
#define CLASSIC_WAVES_PER_SIMD 2
#endif
template <int KS, int S, int RPW, int NT, bool SPLIT>
__global__ __launch_bounds__(256, CLASSIC_WAVES_PER_SIMD) void conv_mfma(const ConvK a) {
    constexpr int BH = 4 * RPW, BW = 32, BN = 32 * NT;
    constexpr int PH = (BH - 1) * S + KS, PW = (BW - 1) * S + KS, PAD = KS / 2;
    constexpr int T = KS * KS;
#ifdef PROBE_TPS3
    constexpr int TPS = KS;
#else
    constexpr int TPS = (KS == 3 && S == 1) ? 9 : KS;  // taps staged in LDS at a time
#endif
    constexpr int NST = T / TPS;
    constexpr int EPI_LD = BN + 4;  // floats per pixel row of the epilogue's transpose tile
    constexpr int LDS_MAIN = PH * PW * LDK + TPS * 4 * BN * 4, LDS_EPI = 4 * 32 * EPI_LD;
    __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI];
    float *patch = lds;
    float *wl = lds + PH * PW * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbn = a.Cout_pad / BN;
    const int nb = blockIdx.x % nbn, tx = blockIdx.x / nbn;
    const int x0 = tx * BW, y0 = blockIdx.y * BH, n0 = nb * BN, img = blockIdx.z;

    f32x16 acc[RPW][NT];
#pragma unroll
    for (int m = 0; m < RPW; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int a_base = ((wave * RPW * S) * PW + (lane & 31) * S) * LDK + (lane >> 5) * 4;
    const int b_base = ((lane >> 5) * BN + (lane & 31)) * 4;

    // ---- software-pipelined main loop.  A "step" is (16-channel chunk, tap stage); while the
    // MFMAs of step k run, the global loads of step k+1 are already in flight into registers
    // (rp: input patch, only when the chunk changes; rw: filter slab) and are written to LDS
    // after the barrier that ends step k.
    constexpr int NP = (PH * PW * 4 + 255) / 256, NW = (TPS * 4 * BN + 255) / 256;
    f32x4 rp[NP], rw[NW];
    struct Cursor {
        int s, c0, cg, st;
    };
    auto advance = [&](Cursor &k) {
        if (++k.st == NST) {
            k.st = 0;
            ++k.cg;
            k.c0 += KC;
            if (k.c0 >= a.seg_C[k.s]) {
                ++k.s;
                k.c0 = 0;
            }
        }
    };
    auto load_patch = [&](const Cursor &k) {
        const int C = a.seg_C[k.s], cs = a.seg_cs[k.s];
        const float *sp = a.seg_ptr[k.s] + (size_t)img * a.Hin * a.Win * cs;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + u * 256;
            const int p = i >> 2, q = i & 3;
            const int py = p / PW, px = p - py * PW;
            const int gy = y0 * S - PAD + py, gx = x0 * S - PAD + px;
            const int c = k.c0 + q * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < PH * PW * 4 && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win && c < C)
                v = *(const f32x4 *)(sp + ((size_t)gy * a.Win + gx) * cs + c);
            rp[u] = v;
        }
    };
    auto store_patch = [&](const Cursor &k) {
        const int C = a.seg_C[k.s];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + u * 256;
            if (i < PH * PW * 4) {
                const int c = k.c0 + (i & 3) * 4;
                f32x4 v = rp[u];
                if (c + 3 >= C) {  // channels past the segment's end read as zero
                    if (c + 1 >= C) v[1] = 0.f;
                    if (c + 2 >= C) v[2] = 0.f;
                    v[3] = 0.f;
                }
                if (a.in_act) {
                    v[0] = act(v[0], a.in_slope);
                    v[1] = act(v[1], a.in_slope);
                    v[2] = act(v[2], a.in_slope);
                    v[3] = act(v[3], a.in_slope);
                }
                if (!SPLIT) {
                    *(f32x4 *)&patch[(i >> 2) * LDK + (i & 3) * 4] = v;
                } else {  // pixel record: [16 x hi fp16 | 16 x lo fp16 | 16 B pad]
                    f16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 h_, l_;
                        split_f16(v[e], h_, l_);
                        hi[e] = h_;
                        lo[e] = l_;
                    }
                    _Float16 *rec = (_Float16 *)&patch[(i >> 2) * LDK];
                    *(f16x4 *)&rec[(i & 3) * 4] = hi;
                    *(f16x4 *)&rec[16 + (i & 3) * 4] = lo;
                }
            }
        }
    };
    auto load_w = [&](const Cursor &k) {
        const float *wsrc = a.wpack + ((size_t)(k.cg * T + k.st * TPS) * 4) * a.Cout_pad * 4 + (size_t)n0 * 4;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * 256;
            const int row = i / BN, col = i - row * BN;
            if (i < TPS * 4 * BN) rw[u] = *(const f32x4 *)(wsrc + ((size_t)row * a.Cout_pad + col) * 4);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * 256;
            if (i < TPS * 4 * BN) *(f32x4 *)&wl[i * 4] = rw[u];
        }
    };

    Cursor cur = {0, 0, 0, 0};
    load_patch(cur);
    load_w(cur);
    while (cur.s < a.nseg) {
        __syncthreads();  // every wave is done reading the previous step's LDS
#ifndef PROBE_NO_STORE
        if (cur.st == 0) store_patch(cur);
        store_w();
#endif
        __syncthreads();
        Cursor nxt = cur;
        advance(nxt);
#ifndef PROBE_NO_LOAD
        if (nxt.s < a.nseg) {
            if (nxt.st == 0) load_patch(nxt);
            load_w(nxt);
        }
#endif
        const int a_st = (TPS == T) ? 0 : cur.st * PW * LDK;  // staged by filter row
#ifdef PROBE_NO_MFMA
        if (a.Hin < 0)
#endif
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int ky = (TPS == T) ? tt / KS : 0, kx = (TPS == T) ? tt % KS : tt;
            if (!SPLIT) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    f32x4 af[RPW], bf[NT];
#pragma unroll
                    for (int m = 0; m < RPW; ++m)
                        af[m] = *(const f32x4 *)&patch[a_base + a_st + ((m * S + ky) * PW + kx) * LDK + k2 * 8];
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        bf[n] = *(const f32x4 *)&wl[b_base + ((tt * 4 + k2 * 2) * BN + n * 32) * 4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < RPW; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][j], bf[n][j], acc[m][n], 0, 0, 0);
                }
            } else {
                // lane half h holds channels 8h..8h+7: hi at float offset 4h, lo at 8 + 4h of the
                // pixel record; filter rows are [tap][hi h0, hi h1, lo h0, lo h1][n][8 fp16]
                f16x8 ah[RPW], al[RPW], bh[NT], bl[NT];
#pragma unroll
                for (int m = 0; m < RPW; ++m) {
                    const float *rec = &patch[a_base + a_st + ((m * S + ky) * PW + kx) * LDK];
                    ah[m] = *(const f16x8 *)rec;
                    al[m] = *(const f16x8 *)(rec + 8);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    bh[n] = *(const f16x8 *)&wl[b_base + ((tt * 4) * BN + n * 32) * 4];
                    bl[n] = *(const f16x8 *)&wl[b_base + ((tt * 4 + 2) * BN + n * 32) * 4];
                }
#pragma unroll
                for (int m = 0; m < RPW; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
                    }
            }
        }
        cur = nxt;
    }

#ifdef PROBE_NO_EPILOGUE
    if (a.Hin > 0 && acc[0][0][0] != 12345.678f) return;
#endif
    // ---- epilogue: bias, activation, (gated) residual(s), NHWC or pixel-shuffled store.
    const int col = lane & 31, hh = lane >> 5;
    const int Cq = a.Cout >> 2;
    const int Cfin = a.ps ? Cq : a.Cout;
    const int Ho = a.ps ? a.Hout * 2 : a.Hout, Wo = a.ps ? a.Wout * 2 : a.Wout;
    const float inv_scale = SPLIT ? 1.f / (ACT_SCALE * WGT_SCALE) : 1.f;
    if (a.vec_epi) {
        // Vector path: the MFMA leaves a channel per lane and pixels in registers; a per-wave
        // transpose through LDS turns that into 4 consecutive channels per lane, so every
        // global access is a 16-byte one (1 KiB per wave-instruction) and all residual loads of
        // a row tile are in flight before the first store (res may alias out: in place is legal,
        // each element is read and written by the same lane).
        constexpr int LPP = BN / 4, PPI = 64 / LPP, NIT = 32 / PPI;
        float *epi = lds + wave * 32 * EPI_LD;
        const int c4 = (lane % LPP) * 4, pl = lane / LPP;
        const int ch = n0 + c4;
        const bool ch_ok = ch < a.Cout;
        int dy = 0, dx = 0, cf = ch;
        if (a.ps) {
            const int sub = ch / Cq;
            cf = ch - sub * Cq;
            dy = sub >> 1;
            dx = sub & 1;
        }
        f32x4 bias = {0.f, 0.f, 0.f, 0.f}, gate = {1.f, 1.f, 1.f, 1.f};
        if (ch_ok) {
            bias = *(const f32x4 *)&a.bpack[ch];
            if (a.res_gate) gate = *(const f32x4 *)&a.res_gate[(size_t)img * Cfin + cf];
        }
        __syncthreads();  // main loop's LDS reads are done in every wave
#pragma unroll
        for (int m = 0; m < RPW; ++m) {
            const int oy = y0 + wave * RPW + m;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    epi[((r & 3) + 8 * (r >> 2) + 4 * hh) * EPI_LD + n * 32 + col] = acc[m][n][r];
            __syncthreads();
            size_t pix[NIT];
            bool ok[NIT];
            f32x4 rv[NIT], rv2[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int ox = x0 + it * PPI + pl;
                ok[it] = ch_ok && oy < a.Hout && ox < a.Wout;
                pix[it] = a.ps ? ((size_t)(img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx) : ((size_t)(img * Ho + oy) * Wo + ox);
                rv[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                rv2[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok[it] && a.res) rv[it] = *(const f32x4 *)&a.res[pix[it] * a.res_cs + cf];
                if (ok[it] && a.res2) rv2[it] = *(const f32x4 *)&a.res2[pix[it] * a.res2_cs + cf];
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                f32x4 v = *(const f32x4 *)&epi[(it * PPI + pl) * EPI_LD + c4];
                v = v * inv_scale + bias;
                if (a.out_act == 1) {
                    v[0] = act(v[0], a.out_slope);
                    v[1] = act(v[1], a.out_slope);
                    v[2] = act(v[2], a.out_slope);
                    v[3] = act(v[3], a.out_slope);
                } else if (a.out_act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), 1.f);
                }
                if (a.res) v = v + (a.res_gate ? rv[it] * gate : rv[it]);
                if (a.res2) v = rv2[it] + v;
                if (ok[it]) *(f32x4 *)&a.out[pix[it] * a.out_cs + cf] = v;
            }
            __syncthreads();
        }
        return;
    }
    // Scalar path (odd channel counts / unaligned slices: 2- and 3-channel outputs).
#pragma unroll
    for (int m = 0; m < RPW; ++m) {
        const int oy = y0 + wave * RPW + m;
        if (oy >= a.Hout) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int ch = n0 + n * 32 + col;
            if (ch >= a.Cout) continue;
            const float bias = a.bpack[ch];
            int dy = 0, dx = 0, cf = ch;
            if (a.ps) {
                const int sub = ch / Cq;
                cf = ch - sub * Cq;
                dy = sub >> 1;
                dx = sub & 1;
            }
            const float gate = a.res_gate ? a.res_gate[(size_t)img * Cfin + cf] : 1.f;
            float rv[16], rv2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // all residual loads first (res may alias out)
                const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const size_t pix = a.ps ? ((size_t)(img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx)
                                        : ((size_t)(img * Ho + oy) * Wo + ox);
                rv[r] = (a.res && ox < a.Wout) ? a.res[pix * a.res_cs + cf] : 0.f;
                rv2[r] = (a.res2 && ox < a.Wout) ? a.res2[pix * a.res2_cs + cf] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (ox >= a.Wout) continue;
                float v = acc[m][n][r] * inv_scale + bias;
                if (a.out_act == 1) v = act(v, a.out_slope);
                else if (a.out_act == 2) v = fminf(fmaxf(v, 0.f), 1.f);
                const size_t pix = a.ps ? ((size_t)(img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx)
                                        : ((size_t)(img * Ho + oy) * Wo + ox);
                if (a.res) v += a.res_gate ? rv[r] * gate : rv[r];
                if (a.res2) v = rv2[r] + v;
                a.out[pix * a.out_cs + cf] = v;
            }
        }
    }
}


// =================================================================================================
// conv_ws: persistent, wave-specialised variant of the same implicit GEMM.
//
// One 512-thread workgroup per CU loops over output tiles (XCD-contiguous ranges, so the 32 CUs
// of an XCD work on neighbouring tiles and share halos / filters in their L2).  Waves 0-3 are
// CONSUMERS: they only read LDS and issue MFMAs (same fragment maps as conv_mfma), then run the
// epilogue.  Waves 4-7 are PRODUCERS: they stage the next step one iteration ahead into the
// other half of double-buffered LDS -- filters by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no
// ds_write), the input patch through registers (it needs the fp32 -> fp16 hi/lo conversion in
// split mode and zero fill at the borders).  One s_barrier per step; because producers run ahead
// across tile boundaries, the consumers' epilogue of tile i overlaps the staging of tile i+1.
//   LDS: 2 x patch + 2 x filter slab + 18 KiB epilogue transpose (3x3: 146.5 KiB, 7x7: 157 KiB).
// =================================================================================================
struct TileCur {
    int i;        // position in this block's tile list
    int x0, y0, n0, img;
    bool valid;
};

struct StepCur {
    TileCur t;
    int s, c0, cg, st;  // segment, first channel of the chunk, global chunk index, tap stage
    bool valid;
};

template <int KS, int S, int RPW, int NT, bool SPLIT>
__global__ __launch_bounds__(512, 2) void conv_ws(const ConvK a, int tiles_x, int tiles_y, int total_tiles) {
    constexpr int BH = 4 * RPW, BW = 32, BN = 32 * NT;
    constexpr int PH = (BH - 1) * S + KS, PW = (BW - 1) * S + KS, PAD = KS / 2;
    constexpr int T = KS * KS;
    constexpr int TPS = (KS == 3 && S == 1) ? 9 : KS;
    constexpr int NST = T / TPS;
    constexpr int PATCH_F = PH * PW * LDK, W_F = TPS * 4 * BN * 4, EPI_LD = 36, EPI_F = 4 * 32 * EPI_LD;
    __shared__ __attribute__((aligned(16))) float lds[2 * PATCH_F + 2 * W_F + EPI_F];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int nbn = a.Cout_pad / BN;

    // ---- tile schedule: XCD x (blockIdx % 8) owns the contiguous tile range [x*R, (x+1)*R)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int R = (total_tiles + 7) >> 3;
    auto tile_at = [&](int i) {
        TileCur t;
        t.i = i;
        int id = xcd * R + i;
        t.valid = i < R && id < total_tiles;
        const int nb = id % nbn;
        id /= nbn;
        const int tx = id % tiles_x;
        id /= tiles_x;
        const int ty = id % tiles_y;
        t.img = id / tiles_y;
        t.x0 = tx * BW;
        t.y0 = ty * BH;
        t.n0 = nb * BN;
        return t;
    };
    auto first_step = [&](const TileCur &t) {
        StepCur k;
        k.t = t;
        k.s = 0;
        k.c0 = 0;
        k.cg = 0;
        k.st = 0;
        k.valid = t.valid;
        return k;
    };
    auto next_chunk = [&](StepCur k) {  // first step of the chunk after k's (possibly in the next tile)
        k.st = 0;
        ++k.cg;
        k.c0 += KC;
        if (k.c0 >= a.seg_C[k.s]) {
            ++k.s;
            k.c0 = 0;
            if (k.s >= a.nseg) k = first_step(tile_at(k.t.i + per_xcd));
        }
        return k;
    };
    auto next_step = [&](StepCur k) {
        if (k.st + 1 < NST) {
            ++k.st;
            return k;
        }
        return next_chunk(k);
    };

    StepCur cur = first_step(tile_at(slot));
    if (!cur.valid) return;  // uniform over the block: no barrier has been executed yet

    if (producer) {
        // ------------------------------------------------------------------ PRODUCER waves
        const int ptid = tid - 256;
        constexpr int NP = (PH * PW * 4 + 255) / 256, NWD = (TPS * 4 * BN + 255) / 256;
        f32x4 rp[NP];
        bool okm[NP];
        auto load_patch = [&](const StepCur &k) {
            // Every lane issues exactly NP loads (out-of-range lanes read the segment's first
            // texel and are zeroed below): the producers' counted s_waitcnt relies on that number.
            const int C = a.seg_C[k.s], cs = a.seg_cs[k.s];
            const float *sp = a.seg_ptr[k.s] + (size_t)k.t.img * a.Hin * a.Win * cs;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int i = ptid + u * 256;
                const int p = i >> 2, q = i & 3;
                const int py = p / PW, px = p - py * PW;
                const int gy = k.t.y0 * S - PAD + py, gx = k.t.x0 * S - PAD + px;
                const int c = k.c0 + q * 4;
                const bool ok = i < PH * PW * 4 && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win && c < C;
                const float *src = ok ? sp + ((size_t)gy * a.Win + gx) * cs + c : sp;
                rp[u] = *(const f32x4 *)src;
                okm[u] = ok;
            }
        };
        auto store_patch = [&](const StepCur &k, float *patch) {
            const int C = a.seg_C[k.s];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int i = ptid + u * 256;
                if (i < PH * PW * 4) {
                    const int c = k.c0 + (i & 3) * 4;
                    f32x4 v = okm[u] ? rp[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (c + 3 >= C) {
                        if (c + 1 >= C) v[1] = 0.f;
                        if (c + 2 >= C) v[2] = 0.f;
                        v[3] = 0.f;
                    }
                    if (a.in_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act(v[e], a.in_slope);
                    }
                    if (!SPLIT) {
                        *(f32x4 *)&patch[(i >> 2) * LDK + (i & 3) * 4] = v;
                    } else {
                        f16x4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            _Float16 h_, l_;
                            split_f16(v[e], h_, l_);
                            hi[e] = h_;
                            lo[e] = l_;
                        }
                        _Float16 *rec = (_Float16 *)&patch[(i >> 2) * LDK];
                        *(f16x4 *)&rec[(i & 3) * 4] = hi;
                        *(f16x4 *)&rec[16 + (i & 3) * 4] = lo;
                    }
                }
            }
        };
        auto dma_w = [&](const StepCur &k, float *wl) {
            const float *wsrc = a.wpack + ((size_t)(k.cg * T + k.st * TPS) * 4) * a.Cout_pad * 4 + (size_t)k.t.n0 * 4;
#pragma unroll
            for (int u = 0; u < NWD; ++u) {
                const int i0 = u * 256 + (wave - 4) * 64;  // wave-uniform: LDS-DMA writes base + lane * 16
                if (i0 < TPS * 4 * BN) {
                    const int i = i0 + lane;
                    const int row = i / BN, col = i - row * BN;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(wsrc + ((size_t)row * a.Cout_pad + col) * 4),
                        (__attribute__((address_space(3))) void *)(wl + i0 * 4), 16, 0, 0);
                }
            }
        };
        // End of a producer iteration: the filter DMA and the ds_writes of this step must have
        // landed; the NP patch loads issued after the DMA (youngest VMEM ops, always exactly NP
        // per lane) stay in flight across the barrier -- they are consumed one iteration later.
        auto sync = [&](bool patch_loads_in_flight) {
            if (patch_loads_in_flight)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

        int g = 0, q = 0;
        StepCur pl = cur;  // the chunk whose patch is (being) held in registers
        load_patch(pl);
        store_patch(pl, lds);
        dma_w(cur, lds + 2 * PATCH_F);
        pl = next_chunk(pl);
        if (pl.valid) load_patch(pl);
        sync(pl.valid);  // barrier #0
        while (true) {
            const StepCur nxt = next_step(cur);
            if (!nxt.valid) {
                __builtin_amdgcn_s_barrier();  // matches the consumers' barrier after the last step
                break;
            }
            bool inflight = false;
            if (nxt.st == 0) {  // a new chunk starts at step g+1: its patch is in rp
                ++q;
                store_patch(nxt, lds + (q & 1) * PATCH_F);
                pl = next_chunk(nxt);
            }
            dma_w(nxt, lds + 2 * PATCH_F + ((g + 1) & 1) * W_F);
            if (nxt.st == 0 && pl.valid) {
                load_patch(pl);
                inflight = true;
            }
            sync(inflight);
            cur = nxt;
            ++g;
        }
        return;
    }

    // ---------------------------------------------------------------------- CONSUMER waves
    const int a_base = ((wave * RPW * S) * PW + (lane & 31) * S) * LDK + (lane >> 5) * 4;
    const int b_base = ((lane >> 5) * BN + (lane & 31)) * 4;
    const int col = lane & 31, hh = lane >> 5;
    const int Cq = a.Cout >> 2;
    const int Cfin = a.ps ? Cq : a.Cout;
    const int Ho = a.ps ? a.Hout * 2 : a.Hout, Wo = a.ps ? a.Wout * 2 : a.Wout;
    const float inv_scale = SPLIT ? 1.f / (ACT_SCALE * WGT_SCALE) : 1.f;
    float *epi = lds + 2 * PATCH_F + 2 * W_F + wave * 32 * EPI_LD;

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // barrier #0: step 0 is staged
    int g = 0, q = 0;
    while (cur.valid) {
        const TileCur tile = cur.t;
        f32x16 acc[RPW][NT];
#pragma unroll
        for (int m = 0; m < RPW; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        // residual tile prefetch: issued before the MFMAs of the tile's LAST step so that the
        // loads' latency is hidden behind them (64 VGPRs for RPW x NT x 4 float4)
        constexpr int LPP = 8, PPI = 8, NIT = 4;  // epilogue geometry: 32-channel halves, 16 B per lane
        const int e_c4 = (lane % LPP) * 4, e_pl = lane / LPP;
        f32x4 rv[RPW][NT][NIT];
        bool rv_loaded = false;
        while (true) {
            const float *patch = lds + (q & 1) * PATCH_F;
            const float *wl = lds + 2 * PATCH_F + (g & 1) * W_F;
            const int a_st = (TPS == T) ? 0 : cur.st * PW * LDK;
            {
                const StepCur peek = next_step(cur);
                if (a.vec_epi && a.res && !(peek.valid && peek.t.i == tile.i)) {
                    rv_loaded = true;
#pragma unroll
                    for (int m = 0; m < RPW; ++m) {
                        const int oy = tile.y0 + wave * RPW + m;
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const int ch = tile.n0 + n * 32 + e_c4;
                            int dy = 0, dx = 0, cf = ch;
                            if (a.ps) {
                                const int sub = ch / Cq;
                                cf = ch - sub * Cq;
                                dy = sub >> 1;
                                dx = sub & 1;
                            }
#pragma unroll
                            for (int it = 0; it < NIT; ++it) {
                                const int ox = tile.x0 + it * PPI + e_pl;
                                const bool ok = ch < a.Cout && oy < a.Hout && ox < a.Wout;
                                const size_t pix = a.ps ? ((size_t)(tile.img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx)
                                                        : ((size_t)(tile.img * Ho + oy) * Wo + ox);
                                rv[m][n][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                                if (ok) rv[m][n][it] = *(const f32x4 *)&a.res[pix * a.res_cs + cf];
                            }
                        }
                    }
                }
            }
            // Software pipeline over "units" (a tap in split mode, half a tap in fp32 mode): the
            // fragments of unit u+1 are requested from LDS before the MFMAs of unit u issue, so a
            // single consumer wave per SIMD never waits on ds_read latency inside a step.
            if (!SPLIT) {
                constexpr int NU = TPS * 2;
                f32x4 af[2][RPW], bf[2][NT];
                auto rd = [&](int u, int sl) {
                    const int tt = u >> 1, k2 = u & 1;
                    const int ky = (TPS == T) ? tt / KS : 0, kx = (TPS == T) ? tt % KS : tt;
#pragma unroll
                    for (int m = 0; m < RPW; ++m)
                        af[sl][m] = *(const f32x4 *)&patch[a_base + a_st + ((m * S + ky) * PW + kx) * LDK + k2 * 8];
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        bf[sl][n] = *(const f32x4 *)&wl[b_base + ((tt * 4 + k2 * 2) * BN + n * 32) * 4];
                };
                rd(0, 0);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (u + 1 < NU) rd(u + 1, (u + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < RPW; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u & 1][m][j], bf[u & 1][n][j], acc[m][n], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                f16x8 ah[2][RPW], al[2][RPW], bh[2][NT], bl[2][NT];
                auto rd = [&](int tt, int sl) {
                    const int ky = (TPS == T) ? tt / KS : 0, kx = (TPS == T) ? tt % KS : tt;
#pragma unroll
                    for (int m = 0; m < RPW; ++m) {
                        const float *rec = &patch[a_base + a_st + ((m * S + ky) * PW + kx) * LDK];
                        ah[sl][m] = *(const f16x8 *)rec;
                        al[sl][m] = *(const f16x8 *)(rec + 8);
                    }
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        bh[sl][n] = *(const f16x8 *)&wl[b_base + ((tt * 4) * BN + n * 32) * 4];
                        bl[sl][n] = *(const f16x8 *)&wl[b_base + ((tt * 4 + 2) * BN + n * 32) * 4];
                    }
                };
                rd(0, 0);
#pragma unroll
                for (int tt = 0; tt < TPS; ++tt) {
                    if (tt + 1 < TPS) rd(tt + 1, (tt + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < RPW; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tt & 1][m], bh[tt & 1][n], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tt & 1][m], bl[tt & 1][n], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tt & 1][m], bh[tt & 1][n], acc[m][n], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's LDS reads are done
            __builtin_amdgcn_s_barrier();
            const StepCur nxt = next_step(cur);
            ++g;
            const bool same_tile = nxt.valid && nxt.t.i == tile.i;
            if (nxt.valid && nxt.st == 0) ++q;
            cur = nxt;
            if (!same_tile) break;
        }

        // ---- epilogue of `tile` (overlaps the producers' staging of the next tile)
        const int x0 = tile.x0, y0 = tile.y0, n0 = tile.n0, img = tile.img;
        if (a.vec_epi) {
            const int c4 = e_c4, pl_ = e_pl;
#pragma unroll
            for (int m = 0; m < RPW; ++m) {
                const int oy = y0 + wave * RPW + m;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int ch = n0 + n * 32 + c4;
                    const bool ch_ok = ch < a.Cout;
                    int dy = 0, dx = 0, cf = ch;
                    if (a.ps) {
                        const int sub = ch / Cq;
                        cf = ch - sub * Cq;
                        dy = sub >> 1;
                        dx = sub & 1;
                    }
                    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, gate = {1.f, 1.f, 1.f, 1.f};
                    if (ch_ok) {
                        bias = *(const f32x4 *)&a.bpack[ch];
                        if (a.res_gate) gate = *(const f32x4 *)&a.res_gate[(size_t)img * Cfin + cf];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) epi[((r & 3) + 8 * (r >> 2) + 4 * hh) * EPI_LD + col] = acc[m][n][r];
                    size_t pix[NIT];
                    bool ok[NIT];
                    f32x4 rv2[NIT];
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int ox = x0 + it * PPI + pl_;
                        ok[it] = ch_ok && oy < a.Hout && ox < a.Wout;
                        pix[it] = a.ps ? ((size_t)(img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx) : ((size_t)(img * Ho + oy) * Wo + ox);
                        rv2[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (ok[it] && a.res2) rv2[it] = *(const f32x4 *)&a.res2[pix[it] * a.res2_cs + cf];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private tile: own ds_writes done
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        f32x4 v = *(const f32x4 *)&epi[(it * PPI + pl_) * EPI_LD + c4];
                        v = v * inv_scale + bias;
                        if (a.out_act == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = act(v[e], a.out_slope);
                        } else if (a.out_act == 2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), 1.f);
                        }
                        if (rv_loaded) v = v + (a.res_gate ? rv[m][n][it] * gate : rv[m][n][it]);
                        if (a.res2) v = rv2[it] + v;
                        if (ok[it]) *(f32x4 *)&a.out[pix[it] * a.out_cs + cf] = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the tile is rewritten
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < RPW; ++m) {
                const int oy = y0 + wave * RPW + m;
                if (oy >= a.Hout) continue;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int ch = n0 + n * 32 + col;
                    if (ch >= a.Cout) continue;
                    const float bias = a.bpack[ch];
                    int dy = 0, dx = 0, cf = ch;
                    if (a.ps) {
                        const int sub = ch / Cq;
                        cf = ch - sub * Cq;
                        dy = sub >> 1;
                        dx = sub & 1;
                    }
                    const float gate = a.res_gate ? a.res_gate[(size_t)img * Cfin + cf] : 1.f;
                    float rv[16], rv2[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const size_t pix = a.ps ? ((size_t)(img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx)
                                                : ((size_t)(img * Ho + oy) * Wo + ox);
                        rv[r] = (a.res && ox < a.Wout) ? a.res[pix * a.res_cs + cf] : 0.f;
                        rv2[r] = (a.res2 && ox < a.Wout) ? a.res2[pix * a.res2_cs + cf] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (ox >= a.Wout) continue;
                        float v = acc[m][n][r] * inv_scale + bias;
                        if (a.out_act == 1) v = act(v, a.out_slope);
                        else if (a.out_act == 2) v = fminf(fmaxf(v, 0.f), 1.f);
                        const size_t pix = a.ps ? ((size_t)(img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx)
                                                : ((size_t)(img * Ho + oy) * Wo + ox);
                        if (a.res) v += a.res_gate ? rv[r] * gate : rv[r];
                        if (a.res2) v = rv2[r] + v;
                        a.out[pix * a.out_cs + cf] = v;
                    }
                }
            }
        }
    }
}

bool use_classic() {
    static const bool v = [] {
        const char *e = getenv("DCVC_CONV_IMPL");
        return !(e && strcmp(e, "ws") == 0);
    }();
    return v;
}

template <int KS, int S, int RPW, int NT>
int launch(const ConvK &k, int N, hipStream_t st, int precision) {
    constexpr int BH = 4 * RPW, BN = 32 * NT;
    if (!use_classic()) {
        const int tiles_x = (k.Wout + 31) / 32, tiles_y = (k.Hout + BH - 1) / BH;
        const int total = N * tiles_y * tiles_x * (k.Cout_pad / BN);
        const int grid = total >= 256 ? 256 : (total + 7) / 8 * 8;  // one persistent workgroup per CU
        if (precision == DCVC_PREC_FP16X3)
            hipLaunchKernelGGL((conv_ws<KS, S, RPW, NT, true>), dim3(grid), dim3(512), 0, st, k, tiles_x, tiles_y, total);
        else
            hipLaunchKernelGGL((conv_ws<KS, S, RPW, NT, false>), dim3(grid), dim3(512), 0, st, k, tiles_x, tiles_y, total);
        return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
    }
    dim3 grid((unsigned)(((k.Wout + 31) / 32) * (k.Cout_pad / BN)), (unsigned)((k.Hout + BH - 1) / BH), (unsigned)N);
    if (precision == DCVC_PREC_FP16X3)
        hipLaunchKernelGGL((conv_mfma<KS, S, RPW, NT, true>), grid, dim3(256), 0, st, k);
    else
        hipLaunchKernelGGL((conv_mfma<KS, S, RPW, NT, false>), grid, dim3(256), 0, st, k);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" int64_t dcvc_conv_pack_size(int32_t Cout, int32_t ks, int32_t nseg, const int32_t *seg_C, int32_t *cout_pad) {
    if (Cout <= 0 || nseg <= 0 || nseg > DCVC_MAX_SEG || (ks != 1 && ks != 3 && ks != 7)) return DCVC_E_ARG;
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) chunks += (seg_C[s] + KC - 1) / KC;
    const int cp = round_up(Cout, 32);
    if (cout_pad) *cout_pad = cp;
    return (int64_t)chunks * ks * ks * 4 * cp * 4;
}

// wpack[((chunk*T + tap)*4 + kq)*Cout_pad + n'][j] = w[n][cin(chunk, kq, j)][tap]
// with n' = n, or for pixel shuffle n' = (n % 4) * (Cout/4) + n / 4 so that the four
// sub-pixel planes are contiguous channel ranges.
extern "C" int dcvc_conv_pack_weights(const float *w, const float *b, int32_t Cout, int32_t ks, int32_t nseg,
                                      const int32_t *seg_C, int32_t pixel_shuffle, int32_t precision, float *wpack,
                                      float *bpack) {
    if (precision != DCVC_PREC_FP32 && precision != DCVC_PREC_FP16X3) return DCVC_E_ARG;
    int32_t cp = 0;
    const int64_t total = dcvc_conv_pack_size(Cout, ks, nseg, seg_C, &cp);
    if (total < 0 || (pixel_shuffle && (Cout & 3))) return DCVC_E_ARG;
    const int T = ks * ks;
    int Cin = 0;
    for (int s = 0; s < nseg; ++s) Cin += seg_C[s];
    memset(wpack, 0, (size_t)total * sizeof(float));
    memset(bpack, 0, (size_t)cp * sizeof(float));
    const int Cq = Cout / 4;
    int cg = 0, cin0 = 0;
    for (int s = 0; s < nseg; ++s) {
        for (int c0 = 0; c0 < seg_C[s]; c0 += KC, ++cg) {
            for (int t = 0; t < T; ++t)
                for (int kq = 0; kq < 4; ++kq)
                    for (int j = 0; j < 4; ++j) {
                        const int c = c0 + kq * 4 + j;
                        if (c >= seg_C[s]) continue;
                        for (int n = 0; n < Cout; ++n) {
                            const int np = pixel_shuffle ? (n & 3) * Cq + (n >> 2) : n;
                            const float v = w[((size_t)n * Cin + cin0 + c) * T + t];
                            if (precision == DCVC_PREC_FP32) {
                                wpack[((((size_t)cg * T + t) * 4 + kq) * cp + np) * 4 + j] = v;
                            } else {
                                // rows [hi h0, hi h1, lo h0, lo h1], 8 fp16 per (row, n): channel 4kq+j = 8h+jj
                                const int cc = kq * 4 + j, h = cc >> 3, jj = cc & 7;
                                float sv = v * WGT_SCALE;
                                sv = sv > F16_MAX ? F16_MAX : (sv < -F16_MAX ? -F16_MAX : sv);
                                const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
                                _Float16 *base = (_Float16 *)wpack;
                                base[((((size_t)cg * T + t) * 4 + h) * cp + np) * 8 + jj] = hi;
                                base[((((size_t)cg * T + t) * 4 + 2 + h) * cp + np) * 8 + jj] = lo;
                            }
                        }
                    }
        }
        cin0 += seg_C[s];
    }
    for (int n = 0; n < Cout; ++n) {
        const int np = pixel_shuffle ? (n & 3) * Cq + (n >> 2) : n;
        bpack[np] = b ? b[n] : 0.f;
    }
    return DCVC_OK;
}

extern "C" int dcvc_conv2d(const dcvc_conv_args *a, void *stream) {
    if (!a || a->nseg < 1 || a->nseg > DCVC_MAX_SEG || !a->out || !a->wpack || !a->bpack) return DCVC_E_ARG;
    if (a->stride != 1 && a->stride != 2) return DCVC_E_ARG;
    if (a->precision != DCVC_PREC_FP32 && a->precision != DCVC_PREC_FP16X3) return DCVC_E_ARG;
    if (a->Cout_pad % 32 || a->Cout > a->Cout_pad || (a->pixel_shuffle && (a->Cout & 3))) return DCVC_E_ARG;
    ConvK k;
    memset(&k, 0, sizeof(k));
    for (int s = 0; s < a->nseg; ++s) {
        if (!a->seg[s].ptr || (a->seg[s].cs & 3) || a->seg[s].cs < round_up(a->seg[s].C, 4) ||
            ((uintptr_t)a->seg[s].ptr & 15))
            return DCVC_E_ARG;
        k.seg_ptr[s] = a->seg[s].ptr;
        k.seg_C[s] = a->seg[s].C;
        k.seg_cs[s] = a->seg[s].cs;
    }
    k.nseg = a->nseg;
    k.Hin = a->Hin;
    k.Win = a->Win;
    const int pad = a->ks / 2;
    k.Hout = (a->Hin + 2 * pad - a->ks) / a->stride + 1;
    k.Wout = (a->Win + 2 * pad - a->ks) / a->stride + 1;
    k.in_act = a->in_act;
    k.in_slope = a->in_slope;
    k.wpack = a->wpack;
    k.bpack = a->bpack;
    k.Cout = a->Cout;
    k.Cout_pad = a->Cout_pad;
    k.out = a->out;
    k.out_cs = a->out_cs;
    k.out_act = a->out_act;
    k.out_slope = a->out_slope;
    k.ps = a->pixel_shuffle;
    k.res = a->res;
    k.res_cs = a->res_cs;
    k.res_gate = a->res_gate;
    k.res2 = a->res2;
    k.res2_cs = a->res2_cs;
    {
        const int cfin = a->pixel_shuffle ? a->Cout / 4 : a->Cout;
        auto al = [](const void *p, int cs) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (cs & 3) == 0); };
        k.vec_epi = (cfin % 4 == 0) && al(a->out, a->out_cs) && al(a->res, a->res_cs) && al(a->res2, a->res2_cs) &&
                    (a->res_gate == nullptr || (((uintptr_t)a->res_gate) & 15) == 0);
    }
    hipStream_t st = (hipStream_t)stream;
    const bool wide = (a->Cout_pad % 64) == 0;
    const int key = a->ks * 10 + a->stride;
    switch (key) {
        case 11: return wide ? launch<1, 1, 2, 2>(k, a->N, st, a->precision) : launch<1, 1, 2, 1>(k, a->N, st, a->precision);
        case 12: return wide ? launch<1, 2, 1, 2>(k, a->N, st, a->precision) : launch<1, 2, 1, 1>(k, a->N, st, a->precision);
        case 31: return wide ? launch<3, 1, 2, 2>(k, a->N, st, a->precision) : launch<3, 1, 2, 1>(k, a->N, st, a->precision);
        case 32: return wide ? launch<3, 2, 1, 2>(k, a->N, st, a->precision) : launch<3, 2, 1, 1>(k, a->N, st, a->precision);
        case 71: return wide ? launch<7, 1, 2, 2>(k, a->N, st, a->precision) : launch<7, 1, 2, 1>(k, a->N, st, a->precision);
        default: return DCVC_E_ARG;
    }
}
