// conv_k32.hip -- split-fp16 stride-1 convolution on v_mfma_f32_16x16x32_f16 with 32-channel K chunks.
//
// Same operator as conv_mfma.hip's DCVC_PREC_FP16X3 path (nn.Conv2d + bias + (Leaky)ReLU + residual(s) + SE gate +
// nn.PixelShuffle(2) + torch.cat prologue: /root/reference/DCVC_HEM/src/layers/layers.py:18-127,
// src/models/video_net.py:74-115,165-223, src/models/video_model.py:17-128), for the layers whose input segments
// are multiples of 32 channels -- every heavy 3x3 layer of the P- and I-picture networks.
//
// Why a second kernel: round 3 measured the bare loops (tools/probes/mfma_loop_probe.hip,
// profiles/r03_mfma_loop_probe.txt): LDS fragment reads + MFMAs only, random operands, equal FLOPs and equal
// 64 px x 64 channel tile per wave.  v_mfma_f32_16x16x32_f16 delivers 1853 TFLOP/s where v_mfma_f32_32x32x16_f16
// delivers 1560: the chip holds 1.87 GHz on the 16x16 shape against 1.62 GHz on the 32x32 one (MI355X_MICROARCH.md
// "DVFS give-back" item 7).  One tap of a 32-channel chunk is exactly one K step of the 16x16x32 instruction.
//
// GEMM view:  Out[pixel][cout] = sum_{chunk, tap, cin in chunk} In[pixel + tap][cin] * W[tap][cin][cout]
//   M = 16 consecutive output pixels of a row, N = 16 output channels, K = 32 input channels of one tap.
// A 256-thread workgroup owns 8 rows x 32 px x BN (= 64 or 32) channels; wave w owns rows 2w, 2w+1 as 4 M tiles
// x BN/16 N tiles (64 accumulator registers at BN = 64).  Per 32-channel chunk the input patch with halo is staged
// once, converted to the split form, and the filter is staged one filter ROW (KS taps) at a time:
//   patch [PH][PW] records of 160 B: [32 x fp16 hi | 32 x fp16 lo | 32 B pad]   (the pad makes the ds_read_b128
//         of 16 consecutive pixels conflict-free: slot = 10 px + kq (mod 16) is a bijection on each lane group)
//   wl    [tap][hi, lo][kq 0..3][BN][8 x fp16]   packed on the host, copied linearly
// = 54 400 + 24 576 B for 3x3 at BN = 64: two workgroups per CU.  Lane l of the MFMA holds A[pixel l & 15]
// [k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][channel l & 15], j = 0..7: channels in natural order.
// x * w ~= xh*wh + xh*wl + xl*wh in that order (hi = fp16(8 x), lo = fp16(8 x - hi); weights scaled by 64), fp32
// accumulate, as in conv_mfma.hip.  Results are deterministic (no atomics on data) but NOT bit-identical to the
// 32x32x16 kernel: the instruction sums 32 products per step instead of 16.  A layer is served by exactly one of the
// two kernels for a given engine configuration, on the encoder and on the decoder side alike.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dcvc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// developer probes (tools/probes/conv_k32_stamps.hip) define K32_STAMP to record s_memtime at phase boundaries;
// in the product library it expands to nothing
#ifndef K32_STAMP
#define K32_STAMP(i)
#endif

namespace {

constexpr int KC = 32;   // channels per K chunk
constexpr int REC = 40;  // floats per pixel record of the LDS patch (160 B)
constexpr float ACT_SCALE = 8.f;
constexpr float WGT_SCALE = 64.f;
constexpr float F16_MAX = 65504.f;
constexpr float ACT_LIMIT = F16_MAX / ACT_SCALE;

struct K32 {
    const float *seg_ptr[DCVC_MAX_SEG];
    int seg_C[DCVC_MAX_SEG];
    int seg_cs[DCVC_MAX_SEG];
    int nseg;
    int H, W;  // stride 1, "same" padding: output size == input size
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
    int *status;
    float *chan_partial;
    int ntx;
    int stagger;  // experiment (DCVC_K32_STAGGER): first-generation workgroups in odd wave slots start this many x 8k cycles late
};

__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int KS, int NTW>
__global__ __launch_bounds__(256, 2) void conv_k32(const K32 a) {
    constexpr int BH = 8, BW = 32, BN = 16 * NTW;
    constexpr int PH = BH + KS - 1, PW = BW + KS - 1, PAD = KS / 2;
    constexpr int T = KS * KS, TPS = KS, NST = KS;  // one filter row of taps in LDS at a time
    constexpr int EPI_LD = BN + 4;
    constexpr int LDS_MAIN = PH * PW * REC + TPS * 8 * BN * 4, LDS_EPI = 4 * 32 * EPI_LD, LDS_RED = 4 * BN;
    __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN > LDS_EPI + LDS_RED ? LDS_MAIN : LDS_EPI + LDS_RED];
    float *patch = lds;
    float *wl = lds + PH * PW * REC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbn = a.Cout_pad / BN;
    const int nb = blockIdx.x % nbn, tx = blockIdx.x / nbn;
    const int x0 = tx * BW, y0 = blockIdx.y * BH, n0 = nb * BN, img = blockIdx.z;

    if (a.stagger) {
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, all bits
        if (lin < 512u && (hwid & 1u))
            for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    f32x4 acc[4][NTW];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // M tile m of a wave: row 2 * wave + (m >> 1), pixels 16 * (m & 1) .. + 15
    const int a_base = ((wave * 2) * PW + (lane & 15)) * REC + (lane >> 4) * 4;
    const int b_base = ((lane >> 4) * BN + (lane & 15)) * 4;

    // ---- software-pipelined main loop: a step is (32-channel chunk, filter row).  While the MFMAs of step k run,
    // the global loads of step k+1 are in flight into registers (rp: the patch, only when the chunk changes; rw: the
    // filter row) and are written to LDS after the barrier that ends step k.
    constexpr int NP = (PH * PW * 8 + 255) / 256, NW = (TPS * 8 * BN + 255) / 256;
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
    int poff[NP];
    unsigned inpic = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int i = tid + u * 256;
        const int p = i >> 3;
        const int py = p / PW, px = p - py * PW;
        const int gy = y0 - PAD + py, gx = x0 - PAD + px;
        const bool ok = i < PH * PW * 8 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        poff[u] = ok ? gy * a.W + gx : 0;
        inpic |= (ok ? 1u : 0u) << u;
    }
    auto load_patch = [&](const Cursor &k) {
        const int cs = a.seg_cs[k.s];
        const float *sp = a.seg_ptr[k.s] + (size_t)img * a.H * a.W * cs + k.c0 + (tid & 7) * 4;
#pragma unroll
        for (int u = 0; u < NP; ++u) rp[u] = *(const f32x4 *)(sp + (size_t)poff[u] * cs);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + u * 256;
            if (i < PH * PW * 8) {
                f32x4 v = ((inpic >> u) & 1u) ? rp[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a.in_act) {
                    v[0] = act(v[0], a.in_slope);
                    v[1] = act(v[1], a.in_slope);
                    v[2] = act(v[2], a.in_slope);
                    v[3] = act(v[3], a.in_slope);
                }
                f32x4 sv = v * ACT_SCALE;
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] = __builtin_amdgcn_fmed3f(sv[e], -F16_MAX, F16_MAX);
                const f16x4 hi = __builtin_convertvector(sv, f16x4);
                const f16x4 lo = __builtin_convertvector(sv - __builtin_convertvector(hi, f32x4), f16x4);
                _Float16 *rec = (_Float16 *)&patch[(i >> 3) * REC];
                *(f16x4 *)&rec[(i & 7) * 4] = hi;
                *(f16x4 *)&rec[32 + (i & 7) * 4] = lo;
            }
        }
    };
    auto load_w = [&](const Cursor &k) {
        const float *wsrc = a.wpack + ((size_t)(k.cg * T + k.st * TPS) * 8) * a.Cout_pad * 4 + (size_t)n0 * 4;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * 256;
            const int row = i / BN, col = i - row * BN;
            if (i < TPS * 8 * BN) rw[u] = *(const f32x4 *)(wsrc + ((size_t)row * a.Cout_pad + col) * 4);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * 256;
            if (i < TPS * 8 * BN) *(f32x4 *)&wl[i * 4] = rw[u];
        }
    };

    auto mfma_step = [&](int st) __attribute__((always_inline)) {
        const int a_st = st * PW * REC;  // filter row ky = st
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl) {  // kx = tl
            f16x8 ah[4], al[4], bh[NTW], bl[NTW];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float *rec = &patch[a_base + a_st + ((m >> 1) * PW + (m & 1) * 16 + tl) * REC];
                ah[m] = *(const f16x8 *)rec;
                al[m] = *(const f16x8 *)(rec + 16);
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                bh[n] = *(const f16x8 *)&wl[b_base + ((tl * 2 + 0) * 4 * BN + n * 16) * 4];
                bl[n] = *(const f16x8 *)&wl[b_base + ((tl * 2 + 1) * 4 * BN + n * 16) * 4];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
    };

    // The patch of chunk c+1 is requested right after chunk c's patch has left the staging registers, a whole chunk
    // (KS steps) ahead of its use instead of one step: the HBM latency of the only operand that comes from beyond L2
    // hides behind three steps of MFMAs (on all-zero operands, i.e. at the full clock, this kernel takes 85 % of its
    // random-data time -- tools/conv_data_probe.py: stalls, not board power, are the larger part of what bounds it).
    Cursor cur = {0, 0, 0, 0};
    K32_STAMP(0);
    load_patch(cur);
    load_w(cur);
    [[maybe_unused]] int stamp_step = 0;
    while (cur.s < a.nseg) {
        K32_STAMP(1 + 4 * stamp_step);
        __syncthreads();  // every wave is done reading the previous step's LDS
        K32_STAMP(2 + 4 * stamp_step);
        if (cur.st == 0) store_patch();
        store_w();
        K32_STAMP(3 + 4 * stamp_step);
        __syncthreads();
        K32_STAMP(4 + 4 * stamp_step);
        ++stamp_step;
        Cursor nxt = cur;
        advance(nxt);
        if (cur.st == 0) {
            Cursor nc = cur;
            nc.c0 += KC;
            if (nc.c0 >= a.seg_C[nc.s]) {
                ++nc.s;
                nc.c0 = 0;
            }
            if (nc.s < a.nseg) load_patch(nc);
        }
        if (nxt.s < a.nseg) load_w(nxt);
        mfma_step(cur.st);
        cur = nxt;
    }

    // ---- epilogue: bias, activation, (gated) residual(s), NHWC or pixel-shuffled store.  The MFMA leaves a channel
    // per lane and 4 pixels in registers; a per-wave transpose through LDS turns that into 4 consecutive channels per
    // lane, so every global access is a 16-byte one, and all residual loads of the wave's rows are in flight before
    // the first store (res may alias out: each element is read and written by the same lane).
    const int Cq = a.Cout >> 2;
    const int Cfin = a.ps ? Cq : a.Cout;
    const int Ho = a.ps ? a.H * 2 : a.H, Wo = a.ps ? a.W * 2 : a.W;
    constexpr float inv_scale = 1.f / (ACT_SCALE * WGT_SCALE);
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
    K32_STAMP(57);
    __syncthreads();  // main loop's LDS reads are done in every wave
    K32_STAMP(58);
    size_t pix[2][NIT];
    bool ok[2][NIT];
    f32x4 rv[2][NIT], rv2[2][NIT];
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};  // this lane's 4 channels summed over its pixels (SE squeeze)
    float vmax = 0.f;                   // largest |output| this lane stores (range guard)
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
        const int oy = y0 + wave * 2 + mr;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int ox = x0 + it * PPI + pl;
            ok[mr][it] = ch_ok && oy < a.H && ox < a.W;
            pix[mr][it] = a.ps ? ((size_t)(img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx) : ((size_t)(img * Ho + oy) * Wo + ox);
            rv[mr][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            rv2[mr][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ok[mr][it] && a.res) rv[mr][it] = *(const f32x4 *)&a.res[pix[mr][it] * a.res_cs + cf];
            if (ok[mr][it] && a.res2) rv2[mr][it] = *(const f32x4 *)&a.res2[pix[mr][it] * a.res2_cs + cf];
        }
    }
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    epi[(hf * 16 + (lane >> 4) * 4 + r) * EPI_LD + n * 16 + (lane & 15)] = acc[mr * 2 + hf][n][r];
        // the transpose tile is private to this wave and a wave's LDS operations execute in order: draining its own
        // ds_writes is all the synchronisation the reads below need
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
            if (a.res) {
                if (a.res_gate) {  // explicit fma: the same rounding in every kernel that applies the SE gate
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rv[mr][it][e], gate[e], v[e]);
                } else {
                    v = v + rv[mr][it];
                }
            }
            if (a.res2) v = rv2[mr][it] + v;
            if (ok[mr][it]) {
                if (a.chan_partial) csum += v;
                // range guard, always on (two v_max3_f32 per 4 outputs): an output beyond +-8188 would be clamped by
                // a split-fp16 consumer.  An infinity is caught here; a NaN can only follow one.
                vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                *(f32x4 *)&a.out[pix[mr][it] * a.out_cs + cf] = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the tile is rewritten
    }
    K32_STAMP(59);
    if (a.status && !(vmax <= ACT_LIMIT)) atomicOr(a.status, DCVC_STATUS_ACT_SATURATED);
    if (a.chan_partial) {
        // SELayer's AdaptiveAvgPool (video_net.py:149-162) rides on the producing convolution: lanes with the same
        // channel quad are LPP apart -> butterfly inside the wave, the four waves through LDS, one partial row per
        // workgroup; dcvc_channel_mean_finish adds the rows in a fixed order (no atomics: encoder and decoder derive
        // bit-identical gates)
#pragma unroll
        for (int off = LPP; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], off);
        float *red = lds + LDS_EPI;
        if (lane < LPP) *(f32x4 *)&red[wave * BN + c4] = csum;
        __syncthreads();
        if (tid < BN && n0 + tid < a.Cout_pad) {
            const float s = ((red[tid] + red[BN + tid]) + red[2 * BN + tid]) + red[3 * BN + tid];
            const size_t part = (size_t)img * (gridDim.y * a.ntx) + (size_t)blockIdx.y * a.ntx + tx;
            a.chan_partial[part * a.Cout_pad + n0 + tid] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv_k32p: the persistent form of the 3x3 kernel above, one 4-wave workgroup per CU with 512 registers per wave.
//
// Why: s_memtime stamps of conv_k32 (tools/conv_k32_stamps.py, profiles/r03_conv_k32_stamps.txt) show a workgroup
// alive for ~64 k cycles per tile of which the matrix pipe needs 13.8 k: the prologue (first loads exposed), the two
// patch conversions (VALU-bound, 2.6 k cycles each with both resident workgroups converting), the epilogue (12-15 k:
// LDS transposes, residual latency, store issue) and the barriers are serial phases that the second resident
// workgroup does not fill (a start stagger changes nothing), and the same picture holds on all-zero operands at the
// full 2.4 GHz clock, i.e. it is the instruction stream, not board power, that bounds the kernel.
//
// Structure: a workgroup walks its tiles as ONE stream of steps (tile, 32-channel chunk, filter row) with every
// operand of step s+1 / chunk q+1 moving while step s computes, across tile boundaries:
//   * two patch buffers and two filter-row buffers in LDS (2 x 54 400 + 2 x 24 576 B): ONE barrier per step;
//   * filter row of step s+2: global -> registers during step s; registers -> LDS at the start of step s+1;
//   * patch of chunk q+1: requested in filter row 0 of chunk q, converted to the split form and written to the other
//     patch buffer in slices between the MFMAs of rows 1 and 2 (VALU work in the shadow of the matrix pipe);
//   * the GEMM is turned around (A = filter, B = pixels): a lane ends up with 4 consecutive output channels of one
//     pixel, so the epilogue needs no LDS transpose and no extra barrier: residual loads are requested before the
//     last filter row's MFMAs, stores are 16 bytes per lane straight from the accumulators;
//   * tiles are dealt in per-XCD bands (workgroups b and b + 8 share an XCD): neighbouring tiles, which share halo
//     rows and columns, run on the same XCD at about the same time and find each other's lines in its L2.
// Same products in the same order as conv_k32 (a x b is commutative, the K order is unchanged): bit-identical.
template <int NTW, bool IN_ACT>
__global__ __launch_bounds__(512) void conv_k32p(const K32 a, const int tiles, const int nty) {
    // 8 waves, two per SIMD: waves w and w + 4 own the same two rows of the tile and one half of its output channels
    // each, so that one wave's VALU / LDS / memory instructions issue beside the other's MFMAs
    constexpr int NTH = 512, NTW2 = NTW / 2;
    static_assert(NTW == 4, "the persistent form is built for 64-channel output blocks");
    constexpr int BH = 8, BW = 32, BN = 16 * NTW, PH = 10, PW = 34, T = 9;
    constexpr int PATCH = PH * PW * REC, WSLAB = 3 * 8 * BN * 4;  // floats
    __shared__ __attribute__((aligned(16))) float lds[2 * PATCH + 2 * WSLAB + 4 * BN];
    float *const pbuf = lds;
    float *const wbuf = lds + 2 * PATCH;
    float *const red = lds + 2 * PATCH + 2 * WSLAB;

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, nh = tid >> 8;  // wave: row pair, nh: channel half
    const int nbn = a.Cout_pad / BN;
    // per-XCD bands of tiles: workgroups b and b + 8 share an XCD
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int band_lo = (int)(((long long)tiles * xcd) >> 3), band_hi = (int)(((long long)tiles * (xcd + 1)) >> 3);

    struct Chunk {  // one 32-channel chunk of one (tile, output-channel block): all fields wave-uniform
        int tile, img, ty, tx;  // tile index and its coordinates (carried incrementally: no divisions in the loop)
        int nb, s, c0, cg;
    };
    auto next_chunk = [&](Chunk c) {
        c.c0 += KC;
        ++c.cg;
        if (c.c0 >= a.seg_C[c.s]) {
            ++c.s;
            c.c0 = 0;
        }
        if (c.s >= a.nseg) {
            c.s = 0;
            c.cg = 0;
            if (++c.nb >= nbn) {
                c.nb = 0;
                c.tile += per;
                c.tx += per;
                while (c.tx >= a.ntx) {
                    c.tx -= a.ntx;
                    if (++c.ty >= nty) {
                        c.ty = 0;
                        ++c.img;
                    }
                }
            }
        }
        return c;
    };
    auto last_of_item = [&](const Chunk &c) { return c.s == a.nseg - 1 && c.c0 + KC >= a.seg_C[c.s]; };

    // ---- staging: patch (11 float4 per thread) and filter row (6 float4 per thread at BN = 64)
    constexpr int NP = (PH * PW * 8 + NTH - 1) / NTH, NW = (3 * 8 * BN + NTH - 1) / NTH;
    static_assert((3 * 8 * BN) % NTH == 0, "filter row is a whole number of float4 per thread");
    f32x4 rp[NP], rw[3][NW];  // rw[r]: filter row r of the chunk that needs it next
    int ppy[NP], ppx[NP];  // patch coordinates of this thread's float4s: tile-independent
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int i = tid + u * NTH;
        const int p = i >> 3;
        ppy[u] = i < PH * PW * 8 ? p / PW : -100000;  // (rows that do not exist never test as inside the picture)
        ppx[u] = p % PW;
    }
    unsigned inpic = 0;  // of the patch held in rp
    auto load_patch = [&](const Chunk &c) __attribute__((always_inline)) {
        const int cs = a.seg_cs[c.s];
        const float *sp = a.seg_ptr[c.s] + (size_t)c.img * a.H * a.W * cs + c.c0 + (tid & 7) * 4;
        const int y0 = c.ty * BH - 1, x0 = c.tx * BW - 1;
        inpic = 0;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int gy = y0 + ppy[u], gx = x0 + ppx[u];
            // branch-free: out-of-picture (and surplus) float4s read pixel 0 and are zeroed when converted
            const unsigned ok = ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
            inpic |= ok << u;
            const int off = ok ? gy * a.W + gx : 0;
            rp[u] = *(const f32x4 *)(sp + (size_t)off * cs);
        }
    };
    // (no control flow inside: a filter-row step must stay ONE basic block for the scheduler to interleave it)
    auto convert_quad = [&](int u, float *patch) __attribute__((always_inline)) {
        const int i = tid + u * NTH;
        f32x4 v = rp[u];
        const float keep = ((inpic >> u) & 1u) ? ACT_SCALE : 0.f;
        if (IN_ACT) {
            v[0] = act(v[0], a.in_slope);
            v[1] = act(v[1], a.in_slope);
            v[2] = act(v[2], a.in_slope);
            v[3] = act(v[3], a.in_slope);
        }
        f32x4 sv = v * keep;  // (x 8, or x 0 outside the picture; loaded values are finite)
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = __builtin_amdgcn_fmed3f(sv[e], -F16_MAX, F16_MAX);
        const f16x4 hi = __builtin_convertvector(sv, f16x4);
        const f16x4 lo = __builtin_convertvector(sv - __builtin_convertvector(hi, f32x4), f16x4);
        // the last quad is partial (2720 float4s, 512 x 6 = 3072 thread-slots): its surplus threads park their zeros in
        // the unused 32-byte pads of the records (two 16-byte slots each) instead of branching around the store
        const bool surplus = u + 1 == NP && i >= PH * PW * 8;
        const int sp = i - PH * PW * 8, srec = sp % (PH * PW), sslot = sp / (PH * PW);
        static_assert(NP * NTH - PH * PW * 8 <= 2 * PH * PW, "pads hold the surplus");
        _Float16 *rec = (_Float16 *)&patch[(surplus ? srec : (i >> 3)) * REC];
        *(f16x4 *)&rec[surplus ? 64 + sslot * 8 : (i & 7) * 4] = hi;
        *(f16x4 *)&rec[surplus ? 68 + sslot * 8 : 32 + (i & 7) * 4] = lo;
    };
    // the NP quads of a patch are converted in six slices, one behind each tap of filter rows 1 and 2
    auto convert_slot = [&](int k, float *patch) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NP; ++u)
            if (u * 6 / NP == k) convert_quad(u, patch);
    };
    auto load_w = [&](const Chunk &c, int ky) __attribute__((always_inline)) {
        const float *wsrc = a.wpack + ((size_t)(c.cg * T + ky * 3) * 8) * a.Cout_pad * 4 + (size_t)(c.nb * BN) * 4;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * NTH;
            const int row = i / BN, col = i - row * BN;
            rw[ky][u] = *(const f32x4 *)(wsrc + ((size_t)row * a.Cout_pad + col) * 4);
        }
    };
    auto store_w = [&](int ky, float *wl) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NW; ++u) *(f32x4 *)&wl[(tid + u * NTH) * 4] = rw[ky][u];
    };
    // s_waitcnt vmcnt(n), n <= 15: everything but the n youngest vector-memory operations has completed.  Explicit, so
    // that the compiler KNOWS the registers requested a step ago are ready: behind the conditional epilogue it cannot
    // count the stores in flight and would make the next use of a loaded register wait for vmcnt(0) -- for the stores
#define K32P_WAIT_ALL_BUT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))
    static_assert(NW + NP <= 15, "count fits the low vmcnt field");

    // ---- accumulators: acc[m][n][r] = output channel n*16 + 4*(lane >> 4) + r of pixel (row 2*wave + (m >> 1),
    // column 16*(m & 1) + (lane & 15)) of the tile
    f32x4 acc[4][NTW2];  // this wave's channel half: n-tiles nh * NTW2 + n
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTW2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    const int a_base = ((wave * 2) * PW + (lane & 15)) * REC + (lane >> 4) * 4;
    const int b_base = ((lane >> 4) * BN + nh * NTW2 * 16 + (lane & 15)) * 4;
    // one tap: 12 fragment reads, 24 MFMAs
    auto mfma_tap = [&](const float *patch, const float *wl, int ky, int tl) __attribute__((always_inline)) {
        f16x8 ah[4], al[4], bh[NTW2], bl[NTW2];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float *rec = &patch[a_base + ((ky + (m >> 1)) * PW + (m & 1) * 16 + tl) * REC];
            ah[m] = *(const f16x8 *)rec;
            al[m] = *(const f16x8 *)(rec + 16);
        }
#pragma unroll
        for (int n = 0; n < NTW2; ++n) {
            bh[n] = *(const f16x8 *)&wl[b_base + ((tl * 2 + 0) * 4 * BN + n * 16) * 4];
            bl[n] = *(const f16x8 *)&wl[b_base + ((tl * 2 + 1) * 4 * BN + n * 16) * 4];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTW2; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], ah[m], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[n], ah[m], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], al[m], acc[m][n], 0, 0, 0);
            }
    };
    // The order asked of the scheduler for one filter-row step (one basic block): the filter-row stores and the
    // global requests first, the first tap's fragments, then per MFMA two VALU slots (the conversion's arithmetic
    // rides in the shadow of the matrix pipe: an MFMA holds the vector issue for 8 of its 16 cycles), the later
    // taps' fragment reads one per three MFMAs, the conversion's LDS stores one per twelve.
#ifndef K32P_SGB
#define K32P_SGB 7  // bit r: ask for the interleave in filter row r's step (developer A/B switch)
#endif
    auto interleave = [&](int row, int nvm, int first) __attribute__((always_inline)) {
        if (!((K32P_SGB >> row) & 1)) return;
        __builtin_amdgcn_sched_group_barrier(0x200, NW, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8 + 2 * NTW2, 0);
        int vm = 0;
#pragma unroll
        for (int i = 0; i < 36 * NTW2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            if (i % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i % 12 == 6) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            // the step's global requests one at a time between the MFMAs (a CU's memory pipeline takes ~24 cycles per
            // 1 KiB wave-request: a burst from all eight waves stalls every wave that touches it for ~1.7 k cycles)
            if (i >= first && (i - first) % 5 == 0 && vm < nvm) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                ++vm;
            }
        }
    };

    // ---- epilogue state
    const int Cq = a.Cout >> 2;
    const int Cfin = a.ps ? Cq : a.Cout;
    const int Ho = a.ps ? a.H * 2 : a.H, Wo = a.ps ? a.W * 2 : a.W;
    constexpr float inv_scale = 1.f / (ACT_SCALE * WGT_SCALE);
    f32x4 rv[4][NTW2];
    float vmax = 0.f;
    unsigned opix[4];   // this lane's pixels of the current item (index before pixel shuffle), 0 when masked
    unsigned ooy[4], oox[4];
    unsigned okm = 0;   // bit m * NTW + n: quad (pixel m, channel quad n) is inside the picture and the layer
    auto out_geo = [&](const Chunk &c) __attribute__((always_inline)) {
        okm = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int oy = c.ty * BH + wave * 2 + (m >> 1), ox = c.tx * BW + (m & 1) * 16 + (lane & 15);
            const bool pok = oy < a.H && ox < a.W;
            ooy[m] = oy;
            oox[m] = ox;
            opix[m] = pok ? (unsigned)((c.img * a.H + oy) * a.W + ox) : 0u;
#pragma unroll
            for (int n = 0; n < NTW2; ++n)
                if (pok && c.nb * BN + (nh * NTW2 + n) * 16 + (lane >> 4) * 4 < a.Cout) okm |= 1u << (m * NTW2 + n);
        }
    };
    // element offset of (pixel m, channel quad starting at ch) in a tensor laid out like the output
    auto out_off = [&](const Chunk &c, int m, int ch, int cs) __attribute__((always_inline)) -> size_t {
        if (!a.ps) return (size_t)opix[m] * cs + ch;
        const int sub = ch / Cq, cf = ch - sub * Cq;
        return ((size_t)(c.img * Ho + 2 * ooy[m] + (sub >> 1)) * Wo + 2 * oox[m] + (sub & 1)) * cs + cf;
    };
    auto load_res = [&](const Chunk &c) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTW2; ++n) {
                const bool ok = (okm >> (m * NTW2 + n)) & 1u;
                const size_t off = out_off(c, m, c.nb * BN + (nh * NTW2 + n) * 16 + (lane >> 4) * 4, a.res_cs);
                rv[m][n] = *(const f32x4 *)&a.res[ok ? off : 0];
            }
    };
    auto epilogue = [&](const Chunk &c) __attribute__((always_inline)) {
        f32x4 csum[NTW2];
#pragma unroll
        for (int n = 0; n < NTW2; ++n) {
            csum[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int ch = c.nb * BN + (nh * NTW2 + n) * 16 + (lane >> 4) * 4;
            const f32x4 bias = *(const f32x4 *)&a.bpack[ch];  // (bpack is Cout_pad long)
            f32x4 gate = {1.f, 1.f, 1.f, 1.f};
            if (a.res_gate && ch < a.Cout) gate = *(const f32x4 *)&a.res_gate[(size_t)c.img * Cfin + (a.ps ? ch % Cq : ch)];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bool ok = (okm >> (m * NTW2 + n)) & 1u;
                f32x4 v = acc[m][n] * inv_scale + bias;
                if (a.out_act == 1) {
                    v[0] = act(v[0], a.out_slope);
                    v[1] = act(v[1], a.out_slope);
                    v[2] = act(v[2], a.out_slope);
                    v[3] = act(v[3], a.out_slope);
                } else if (a.out_act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), 1.f);
                }
                if (a.res) {
                    if (a.res_gate) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rv[m][n][e], gate[e], v[e]);
                    } else {
                        v = v + rv[m][n];
                    }
                }
                if (ok) {
                    if (a.res2) v = *(const f32x4 *)&a.res2[out_off(c, m, ch, a.res2_cs)] + v;
                    if (a.chan_partial) csum[n] += v;
                    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    *(f32x4 *)&a.out[out_off(c, m, ch, a.out_cs)] = v;
                }
            }
        }
        if (a.chan_partial) {
            // SE squeeze: sum over the 16 pixel lanes of a channel quad (butterfly), then the four waves through LDS;
            // fixed order -> the same bits on the encoder and the decoder side
#pragma unroll
            for (int n = 0; n < NTW2; ++n)
#pragma unroll
                for (int off = 1; off < 16; off <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[n][e] += __shfl_xor(csum[n][e], off);
            if ((lane & 15) == 0)
#pragma unroll
                for (int n = 0; n < NTW2; ++n) *(f32x4 *)&red[wave * BN + (nh * NTW2 + n) * 16 + (lane >> 4) * 4] = csum[n];
            __syncthreads();
            if (tid < BN && c.nb * BN + tid < a.Cout_pad) {
                const float s = ((red[tid] + red[BN + tid]) + red[2 * BN + tid]) + red[3 * BN + tid];
                a.chan_partial[(size_t)c.tile * a.Cout_pad + c.nb * BN + tid] = s;
            }
            __syncthreads();
        }
    };

    // ---- pipeline prologue (once per workgroup, not per tile)
    Chunk cq;
    cq.tile = band_lo + (int)(blockIdx.x >> 3);
    if (cq.tile >= band_hi) return;
    {
        const int per_img = nty * a.ntx;
        cq.img = cq.tile / per_img;
        const int r = cq.tile - cq.img * per_img;
        cq.ty = r / a.ntx;
        cq.tx = r - cq.ty * a.ntx;
        cq.nb = cq.s = cq.c0 = cq.cg = 0;
    }
    // past the end of the workgroup's stream the look-ahead repeats the last chunk (harmless re-reads): no branches
    auto next_or_same = [&](const Chunk &c) {
        const Chunk n = next_chunk(c);
        return n.tile < band_hi ? n : c;
    };
    load_patch(cq);
    load_w(cq, 0);
#pragma unroll
    for (int u = 0; u < NP; ++u) convert_quad(u, pbuf);
    store_w(0, wbuf);
    load_w(cq, 1);
    load_w(cq, 2);
    Chunk cn = next_or_same(cq);
    load_patch(cn);
    out_geo(cq);
    K32P_WAIT_ALL_BUT(0);
    __syncthreads();
    int pq = 0, par = 0;  // patch / filter-row buffer in use

    // Per step: the filter row of the NEXT step goes registers -> LDS at the start; the global requests (the filter
    // row three steps ahead, after row 2 the patch two chunks ahead) go out at the END, behind the MFMAs and in front
    // of the epilogue's stores, so that a burst of stores is never in front of a request in the CU's memory queue
    // (stamped build: 4-6 k cycles per tile went there); every request has a full step before its data is touched.
    for (;;) {
        const bool more = next_chunk(cq).tile < band_hi;
        const Chunk cnn = next_or_same(cn);
        const bool last = last_of_item(cq);
        float *const patch = pbuf + pq * PATCH, *const npatch = pbuf + (pq ^ 1) * PATCH;
        [[maybe_unused]] const bool stamp_on = cq.tile == band_lo + (int)(blockIdx.x >> 3) + 2 * per && cq.cg == 0 && cq.nb == 0;
#define K32P_STAMP(i) \
    if (stamp_on) K32_STAMP(i)
        // -------- filter row 0
        {
            K32P_STAMP(0);
            float *const wl = wbuf + par * WSLAB;
            store_w(1, wbuf + (par ^ 1) * WSLAB);
            load_w(cn, 0);
#pragma unroll
            for (int tl = 0; tl < 3; ++tl) mfma_tap(patch, wl, 0, tl);
            interleave(0, NW, 40);  // late in the step: the previous item's stores may still be draining
            K32P_WAIT_ALL_BUT(NW);
            K32P_STAMP(2);
            __syncthreads();
            par ^= 1;
        }
        // -------- filter row 1: first half of the next patch's conversion rides between the MFMAs
        {
            K32P_STAMP(3);
            float *const wl = wbuf + par * WSLAB;
            store_w(2, wbuf + (par ^ 1) * WSLAB);
            load_w(cn, 1);
#pragma unroll
            for (int tl = 0; tl < 3; ++tl) {
                mfma_tap(patch, wl, 1, tl);
                convert_slot(tl, npatch);
            }
            interleave(1, NW, 8);
            K32P_WAIT_ALL_BUT(NW);
            K32P_STAMP(5);
            __syncthreads();
            par ^= 1;
        }
        // -------- filter row 2: second half of the conversion; the item's epilogue after its last chunk
        {
            K32P_STAMP(6);
            float *const wl = wbuf + par * WSLAB;
            store_w(0, wbuf + (par ^ 1) * WSLAB);  // row 0 of the next chunk
            if (last && a.res) load_res(cq);
            load_w(cn, 2);
#pragma unroll
            for (int tl = 0; tl < 3; ++tl) {
                mfma_tap(patch, wl, 2, tl);
                convert_slot(3 + tl, npatch);
            }
            load_patch(cnn);  // (after the last conversion in program order: the staging registers are free)
            interleave(2, NW + NP, 4);
            K32P_WAIT_ALL_BUT(NW + NP);
            K32P_STAMP(8);
            if (last) {
                epilogue(cq);
                zero_acc();
                if (more) out_geo(cn);
            }
            K32P_STAMP(9);
            __syncthreads();
            K32P_STAMP(10);
            par ^= 1;
        }
        if (!more) break;
        cq = cn;
        cn = cnn;
        pq ^= 1;
    }
    if (a.status && !(vmax <= ACT_LIMIT)) atomicOr(a.status, DCVC_STATUS_ACT_SATURATED);
}

template <int NTW>
int launch_persistent(K32 &k, int N, hipStream_t st) {
    k.ntx = (k.W + 31) / 32;
    const int nty = (k.H + 7) / 8, tiles = N * nty * k.ntx;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return DCVC_E_LAUNCH;
        cus = prop.multiProcessorCount / 8 * 8;
    }
    if (k.in_act)
        hipLaunchKernelGGL((conv_k32p<NTW, true>), dim3((unsigned)cus), dim3(512), 0, st, k, tiles, nty);
    else
        hipLaunchKernelGGL((conv_k32p<NTW, false>), dim3((unsigned)cus), dim3(512), 0, st, k, tiles, nty);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}

template <int KS, int NTW>
int launch(K32 &k, int N, hipStream_t st) {
    constexpr int BN = 16 * NTW;
    k.ntx = (k.W + 31) / 32;
    dim3 grid((unsigned)(k.ntx * (k.Cout_pad / BN)), (unsigned)((k.H + 7) / 8), (unsigned)N);
    hipLaunchKernelGGL((conv_k32<KS, NTW>), grid, dim3(256), 0, st, k);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline bool aligned16(const void *p, int cs) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (cs & 3) == 0); }

}  // namespace

extern "C" int64_t dcvc_conv_k32_pack_bytes(int32_t Cout, int32_t ks, int32_t nseg, const int32_t *seg_C, int32_t *cout_pad) {
    if (Cout <= 0 || nseg <= 0 || nseg > DCVC_MAX_SEG || (ks != 1 && ks != 3)) return DCVC_E_ARG;
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) {
        if (seg_C[s] <= 0 || seg_C[s] % KC) return DCVC_E_ARG;
        chunks += seg_C[s] / KC;
    }
    const int cp = round_up(Cout, 32);
    if (cout_pad) *cout_pad = cp;
    return (int64_t)chunks * ks * ks * 8 * cp * 16;
}

// wpack as 16-byte entries: entry[((chunk * T + tap) * 2 + hl) * 4 + kq][n'] = 8 fp16: w[n][cin = 32 chunk + 8 kq + j][tap]
// (hl 0: hi, 1: lo of 64 w), n' = n, or with pixel shuffle n' = (n % 4) * (Cout / 4) + n / 4 (sub-pixel planes contiguous).
extern "C" int dcvc_conv_k32_pack_weights(const float *w, const float *b, int32_t Cout, int32_t ks, int32_t nseg,
                                          const int32_t *seg_C, int32_t pixel_shuffle, void *wpack, float *bpack) {
    int32_t cp = 0;
    const int64_t total = dcvc_conv_k32_pack_bytes(Cout, ks, nseg, seg_C, &cp);
    if (total < 0 || (pixel_shuffle && (Cout & 3))) return DCVC_E_ARG;
    const int T = ks * ks;
    int Cin = 0;
    for (int s = 0; s < nseg; ++s) Cin += seg_C[s];
    memset(wpack, 0, (size_t)total);
    memset(bpack, 0, (size_t)cp * sizeof(float));
    _Float16 *base = (_Float16 *)wpack;
    const int Cq = Cout / 4;
    bool clamped = false;
    for (int c = 0; c < Cin; ++c) {  // segments are multiples of 32: chunk boundaries never straddle one
        const int cg = c / KC, kq = (c % KC) >> 3, j = c & 7;
        for (int t = 0; t < T; ++t)
            for (int n = 0; n < Cout; ++n) {
                const int np = pixel_shuffle ? (n & 3) * Cq + (n >> 2) : n;
                float sv = w[((size_t)n * Cin + c) * T + t] * WGT_SCALE;
                if (!(fabsf(sv) <= F16_MAX)) {
                    clamped = true;
                    sv = sv > 0.f ? F16_MAX : -F16_MAX;
                }
                const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
                base[(((((size_t)cg * T + t) * 2 + 0) * 4 + kq) * cp + np) * 8 + j] = hi;
                base[(((((size_t)cg * T + t) * 2 + 1) * 4 + kq) * cp + np) * 8 + j] = lo;
            }
    }
    for (int n = 0; n < Cout; ++n) {
        const int np = pixel_shuffle ? (n & 3) * Cq + (n >> 2) : n;
        bpack[np] = b ? b[n] : 0.f;
    }
    return clamped ? DCVC_E_RANGE : DCVC_OK;
}

extern "C" int dcvc_conv2d_k32(const dcvc_conv_args *a, void *stream) {
    if (!a || a->nseg < 1 || a->nseg > DCVC_MAX_SEG || !a->out || !a->wpack || !a->bpack) return DCVC_E_ARG;
    if (a->stride != 1 || (a->ks != 1 && a->ks != 3) || a->precision != DCVC_PREC_FP16X3) return DCVC_E_ARG;
    if (a->Cout_pad % 32 || a->Cout > a->Cout_pad || (a->pixel_shuffle && (a->Cout & 3))) return DCVC_E_ARG;
    K32 k;
    memset(&k, 0, sizeof(k));
    for (int s = 0; s < a->nseg; ++s) {
        if (!a->seg[s].ptr || a->seg[s].C <= 0 || a->seg[s].C % KC || (a->seg[s].cs & 3) || a->seg[s].cs < a->seg[s].C ||
            ((uintptr_t)a->seg[s].ptr & 15))
            return DCVC_E_ARG;
        k.seg_ptr[s] = a->seg[s].ptr;
        k.seg_C[s] = a->seg[s].C;
        k.seg_cs[s] = a->seg[s].cs;
    }
    k.nseg = a->nseg;
    k.H = a->Hin;
    k.W = a->Win;
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
    k.status = a->status;
    k.chan_partial = a->chan_partial;
    static const int stagger = getenv("DCVC_K32_STAGGER") ? atoi(getenv("DCVC_K32_STAGGER")) : 0;
    k.stagger = stagger;
    const int cfin = a->pixel_shuffle ? a->Cout / 4 : a->Cout;
    // this kernel has the 16-byte epilogue only (every layer it is meant for qualifies); others stay on dcvc_conv2d
    if ((cfin % 4) || !aligned16(a->out, a->out_cs) || !aligned16(a->res, a->res_cs) || !aligned16(a->res2, a->res2_cs) ||
        (a->res_gate && (((uintptr_t)a->res_gate) & 15)))
        return DCVC_E_ARG;
    if (a->chan_partial && a->pixel_shuffle) return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = (a->Cout_pad % 64) == 0;
    if (a->ks == 3) {
        // the persistent form needs enough tiles to give every CU a few; small pictures stay on the simple kernel
        // (DCVC_K32_PERSISTENT: 0 never, 1 by size (default), 2 always -- developer / test switch, read per call)
        const char *env = getenv("DCVC_K32_PERSISTENT");
        const int mode = env ? atoi(env) : 1;
        const long long tiles = (long long)a->N * ((a->Hin + 7) / 8) * ((a->Win + 31) / 32);
        if (wide && (mode == 2 || (mode == 1 && tiles >= 512))) return launch_persistent<4>(k, a->N, st);
        return wide ? launch<3, 4>(k, a->N, st) : launch<3, 2>(k, a->N, st);
    }
    return wide ? launch<1, 4>(k, a->N, st) : launch<1, 2>(k, a->N, st);
}
