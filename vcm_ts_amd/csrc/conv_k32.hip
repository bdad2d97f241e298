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

// (Developer builds -- s_memtime stamps at the phase boundaries, ablation variants without MFMAs / fragment reads /
// stores, the staging and epilogue alternatives measured in round 3 -- are produced by applying
// tools/probes/conv_k32_dev_switches.patch to a COPY of this file: tools/probes/Makefile, tools/build_variant.sh.)

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
    int ty0, nty, band_rows;  // band of tile rows this launch computes (dcvc_conv_args.tile_row0 / tile_rows)
};

__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int KS, int NTW, int NWAVE>
__global__ __launch_bounds__(64 * NWAVE, NWAVE / 2) void conv_k32(const K32 a) {
    // NWAVE waves per workgroup (4 or 8): wave w owns RW = 8 / NWAVE rows of the tile, i.e. MT = 2 * RW M tiles.  With 8
    // waves two resident workgroups put FOUR waves on every SIMD (128 registers each) instead of two
    constexpr int NTH = 64 * NWAVE, RW = 8 / NWAVE, MT = 2 * RW;

    constexpr int BH = 8, BW = 32, BN = 16 * NTW;
    constexpr int PH = BH + KS - 1, PW = BW + KS - 1, PAD = KS / 2;
    constexpr int T = KS * KS, TPS = KS, NST = KS;  // one filter row of taps in LDS at a time
    constexpr int LDS_MAIN = PH * PW * REC + TPS * 8 * BN * 4;
    static_assert(LDS_MAIN >= 8 * BN, "the SE reduction reuses the front of the buffer");
    __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN];
    float *patch = lds;
    float *wl = lds + PH * PW * REC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbn = a.Cout_pad / BN;
    const int nb = blockIdx.x % nbn, tx = blockIdx.x / nbn;
    const int ty = blockIdx.y + a.ty0;
    const int x0 = tx * BW, y0 = ty * BH, n0 = nb * BN, img = blockIdx.z;

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // M tile m of a wave: row RW * wave + (m >> 1), pixels 16 * (m & 1) .. + 15
    const int a_base = ((wave * RW) * PW + (lane & 15)) * REC + (lane >> 4) * 4;
    const int b_base = ((lane >> 4) * BN + (lane & 15)) * 4;

    // ---- software-pipelined main loop: a step is (32-channel chunk, filter row).  While the MFMAs of step k run,
    // the global loads of step k+1 are in flight into registers (rp: the patch, only when the chunk changes; rw: the
    // filter row) and are written to LDS after the barrier that ends step k.
    constexpr int NP = (PH * PW * 8 + NTH - 1) / NTH, NW = (TPS * 8 * BN + NTH - 1) / NTH;
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
    // Nothing about a lane's staging slots is kept in registers across the loop (round 3: the 8-wave build had no room for
    // the 6 pixel offsets + 3 filter offsets and spilled two of them to scratch, whose reloads -- vector-memory loads
    // with a vmcnt(0) each -- sat in front of every chunk's patch request): a slot's patch pixel and its filter offset
    // are recomputed where they are used, a dozen VALU operations per 16-byte load.
    // Address arithmetic is kept off the 64-bit vector ALU: every global access is a wave-uniform base pointer (scalar
    // registers) plus a 32-bit per-lane byte offset (tensors stay below 4 GiB).
    static_assert((PH * PW * 241 < (1 << 23) && PW == 34) || KS != 3, "the multiply-shift below divides by PW = 34");
    auto ld16 = [](const void *base, unsigned byte_off) __attribute__((always_inline)) {
        return *(const f32x4 *)((const char *)base + byte_off);
    };
    // Staging slot -> patch pixel: the middle two of every four consecutive slots are swapped (0 2 1 3).  A ds_write_b64 is
    // served in groups of 16 lanes = two pixels' 64-byte halves; with neighbouring pixels (records 160 B = 40 banks apart,
    // stores banked modulo 32) those overlap on 8 banks -- the 7 % bank-conflict cycles of profiles/r03_..._sq_counters.txt
    // -- while pixels two apart (80 banks = 16 modulo 32) do not.  Which lane stages which pixel changes no value.
    static_assert((PH * PW) % 4 == 0, "slot_pixel permutes inside groups of four pixels");
    auto slot_pixel = [](int q) __attribute__((always_inline)) { return (q & ~3) | ((q & 1) << 1) | ((q >> 1) & 1); };
    auto load_patch = [&](const Cursor &k) {
        const unsigned cs4 = (unsigned)a.seg_cs[k.s] * 4u;  // bytes per pixel of this segment (< 2^24)
        const char *sp = (const char *)(a.seg_ptr[k.s] + (size_t)img * a.H * a.W * a.seg_cs[k.s] + k.c0);
        int t = tid;
        asm volatile("" : "+v"(t));  // (opaque: keeps the compiler from hoisting the slot arithmetic back out of the loop)
        const unsigned lane_off = (t & 7) * 16u;
        // the image as a raw buffer: an offset at or beyond num_records reads as zeros (no select per value for the
        // out-of-picture slots of the halo)
        const unsigned img_bytes = (unsigned)(a.H * a.W) * cs4;  // (< 2^32: dcvc_conv2d_k32 checks)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)sp, 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = t + u * NTH;
            const int p = slot_pixel(i >> 3);
            const int py = PW == 34 ? (p * 241) >> 13 : p / PW, px = p - py * PW;
            const int gy = y0 - PAD + py, gx = x0 - PAD + px;
            // (bitwise on purpose: && would turn every slot into a branch)
            const unsigned ok = (unsigned)(tid + u * NTH < PH * PW * 8) & (unsigned)((unsigned)gy < (unsigned)a.H) &
                                (unsigned)((unsigned)gx < (unsigned)a.W);
            const unsigned off = ok ? __umul24((unsigned)(gy * a.W + gx), cs4) + lane_off : img_bytes;
            rp[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + u * NTH;
            if (i < PH * PW * 8) {
                // LeakyReLU on the scaled value: max(s, s * slope) == 8 * (v > 0 ? v : v * slope) for 0 <= slope <= 1
                // (a power-of-two scale commutes with the rounding of v * slope; dcvc_conv2d_k32 checks the slope)
                f32x4 sv = rp[u] * ACT_SCALE;
                if (a.in_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {  // (one v_max_f32: fmaxf would add a NaN-quieting operation per value)
                        const float t = sv[e] * a.in_slope;
                        float r;
                        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(sv[e]), "v"(t));
                        sv[e] = r;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] = __builtin_amdgcn_fmed3f(sv[e], -F16_MAX, F16_MAX);
                const f16x4 hi = __builtin_convertvector(sv, f16x4);
                const f16x4 lo = __builtin_convertvector(sv - __builtin_convertvector(hi, f32x4), f16x4);
                _Float16 *rec = (_Float16 *)&patch[slot_pixel(i >> 3) * REC];
                *(f16x4 *)&rec[(i & 7) * 4] = hi;
                *(f16x4 *)&rec[32 + (i & 7) * 4] = lo;
            }
        }
    };
    auto load_w = [&](const Cursor &k) {
        const char *wsrc = (const char *)(a.wpack + ((size_t)(k.cg * T + k.st * TPS) * 8) * a.Cout_pad * 4 + (size_t)n0 * 4);
        int t = tid;
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = t + u * NTH;  // float4 i of the filter row: [row = i / BN][col = i % BN], rows Cout_pad apart
            if (u + 1 < NW || tid + u * NTH < TPS * 8 * BN) rw[u] = ld16(wsrc, (unsigned)((i / BN) * a.Cout_pad + (i % BN)) * 16u);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * NTH;
            if (i < TPS * 8 * BN) *(f32x4 *)&wl[i * 4] = rw[u];
        }
    };

    auto mfma_step = [&](int st) __attribute__((always_inline)) {
        const int a_st = st * PW * REC;  // filter row ky = st
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl) {  // kx = tl
            f16x8 ah[MT], al[MT], bh[NTW], bl[NTW];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
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
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    // filter as the A operand, pixels as B: D[channel][pixel], i.e. a lane ends up with 4 CONSECUTIVE
                    // channels (n*16 + 4*(lane >> 4) + r) of one pixel (lane & 15 of M tile m): 16-byte epilogue accesses
                    // without a transpose.  Same products in the same K order as the other orientation.
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], ah[m], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[n], ah[m], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], al[m], acc[m][n], 0, 0, 0);
                }
        }
    };

    // The patch of chunk c+1 is requested right after chunk c's patch has left the staging registers, a whole chunk
    // (KS steps) ahead of its use instead of one step: the HBM latency of the only operand that comes from beyond L2
    // hides behind three steps of MFMAs (on all-zero operands, i.e. at the full clock, this kernel takes 85 % of its
    // random-data time -- tools/conv_data_probe.py: stalls, not board power, are the larger part of what bounds it).
    Cursor cur = {0, 0, 0, 0};
    load_patch(cur);
    load_w(cur);
    while (cur.s < a.nseg) {
        __syncthreads();  // every wave is done reading the previous step's LDS
        if (cur.st == 0) store_patch();
        store_w();
        __syncthreads();
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

    // ---- epilogue: bias, activation, (gated) residual(s), NHWC or pixel-shuffled store, straight from the
    // accumulators: lane l holds, for M tile m and N tile n, the channel quad n0 + n*16 + 4*(l >> 4) of pixel
    // (row 2*wave + (m >> 1), column 16*(m & 1) + (l & 15)).  No LDS, no barrier: a wave starts its epilogue when ITS
    // last MFMA is done.  All residual requests go out before the first store (res may alias out: each element is
    // read and written by the same lane).  (The first version transposed through LDS as conv_mfma does: 12-15 k of a
    // workgroup's 64 k cycles, profiles/r03_conv_k32_stamps.txt.)
    const int Cq = a.Cout >> 2;
    const int Cfin = a.ps ? Cq : a.Cout;
    const int Ho = a.ps ? a.H * 2 : a.H, Wo = a.ps ? a.W * 2 : a.W;
    constexpr float inv_scale = 1.f / (ACT_SCALE * WGT_SCALE);
    int oy[MT], ox[MT];
    unsigned okm = 0;  // bit m * NTW + n
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        oy[m] = y0 + wave * RW + (m >> 1);
        ox[m] = x0 + (m & 1) * 16 + (lane & 15);
#pragma unroll
        for (int n = 0; n < NTW; ++n)
            if (oy[m] < a.H && ox[m] < a.W && n0 + n * 16 + (lane >> 4) * 4 < a.Cout) okm |= 1u << (m * NTW + n);
    }
    // byte offset of (pixel m, channel quad n) in a tensor laid out like the output, from the image's base
    // (32-bit: an image of a tensor stays below 4 GiB)
    unsigned pixo[MT];  // pixel index inside the (pixel-shuffled) output image (< 2^24)
#pragma unroll
    for (int m = 0; m < MT; ++m) pixo[m] = a.ps ? (unsigned)((2 * oy[m]) * Wo + 2 * ox[m]) : (unsigned)(oy[m] * a.W + ox[m]);
    unsigned chq[NTW], pso[NTW];  // per channel quad: channel inside its (sub-pixel) plane, pixel offset of that plane
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int ch = n0 + n * 16 + (lane >> 4) * 4;
        const int sub = a.ps ? ch / Cq : 0;  // (Cq % 4 == 0: a quad stays inside one sub-pixel plane)
        chq[n] = (unsigned)(ch - sub * Cq);
        pso[n] = (unsigned)((sub >> 1) * Wo + (sub & 1));
    }
    // (24-bit multiplies: full-rate VALU; v_mul_lo_u32 is quarter rate)
    auto out_off = [&](int m, int n, int cs) __attribute__((always_inline)) -> unsigned {
        return (__umul24(pixo[m] + pso[n], (unsigned)cs) + chq[n]) * 4u;
    };
    const size_t img_pix = (size_t)img * Ho * Wo;  // (uniform)
    const char *res_b = a.res ? (const char *)(a.res + img_pix * a.res_cs) : nullptr;
    const char *res2_b = a.res2 ? (const char *)(a.res2 + img_pix * a.res2_cs) : nullptr;
    char *out_b = (char *)(a.out + img_pix * a.out_cs);
    f32x4 rv[MT][NTW];
    // the output / residual images as raw buffers (scalar base + 32-bit offset; a masked lane gets offset == num_records:
    // its load returns zeros, its store is dropped -- no branch per access)
    const unsigned out_bytes = (unsigned)(Ho * Wo) * (unsigned)a.out_cs * 4u, res_bytes = (unsigned)(Ho * Wo) * (unsigned)a.res_cs * 4u;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)out_b, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)res_b, 0, a.res ? (int)res_bytes : 0, 0x00020000);
    if (a.res) {  // (requesting these under the last step's MFMAs was tried, also with the last step peeled off the loop:
                  // the compiler keeps or spills their 32-64 registers; requesting only their cache lines there (LDS-DMA
                  // loads into a sink) made the launch 3 % slower, tools/ab_probe.py)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const bool ok = (okm >> (m * NTW + n)) & 1u;
                rv[m][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)(ok ? out_off(m, n, a.res_cs) : res_bytes), 0, 0));
            }
    }
    // (fused SE squeeze) channel sums per ROW of the tile, so that the result does not depend on how many rows a wave owns
    f32x4 csum[RW][NTW];
    float vmax = 0.f;  // largest |output| this lane stores (range guard)
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
#pragma unroll
        for (int r = 0; r < RW; ++r) csum[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ch = n0 + n * 16 + (lane >> 4) * 4;
        const f32x4 bias = *(const f32x4 *)&a.bpack[ch];  // (bpack is Cout_pad long)
        f32x4 gate = {1.f, 1.f, 1.f, 1.f};
        if (a.res_gate && ch < a.Cout) gate = *(const f32x4 *)&a.res_gate[(size_t)img * Cfin + (a.ps ? ch % Cq : ch)];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
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
                if (a.res_gate) {  // explicit fma: the same rounding in every kernel that applies the SE gate
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rv[m][n][e], gate[e], v[e]);
                } else {
                    v = v + rv[m][n];
                }
            }
            const bool ok = (okm >> (m * NTW + n)) & 1u;
            if (!a.res2 && !a.chan_partial) {
                // range guard, always on (two v_max3_f32 per 4 outputs): an output beyond +-8188 would be clamped by
                // a split-fp16 consumer.  An infinity is caught here; a NaN can only follow one.
                const float m4 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                vmax = fmaxf(vmax, ok ? m4 : 0.f);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) int, v), rs_out,
                                                       (int)(ok ? out_off(m, n, a.out_cs) : out_bytes), 0, 0);
            } else if (ok) {
                if (a.res2) v = ld16(res2_b, out_off(m, n, a.res2_cs)) + v;
                if (a.chan_partial) csum[m >> 1][n] += v;
                vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                *(f32x4 *)(out_b + out_off(m, n, a.out_cs)) = v;
            }
        }
    }
    if (a.status && !(vmax <= ACT_LIMIT)) atomicOr(a.status, DCVC_STATUS_ACT_SATURATED);
    if (a.chan_partial) {
        // SELayer's AdaptiveAvgPool (video_net.py:149-162) rides on the producing convolution: per tile ROW the two
        // 16-pixel halves, then the 16 pixel lanes of a channel quad by a butterfly inside the wave, the 8 rows through
        // LDS in row order, one partial row per workgroup; dcvc_channel_mean_finish adds those in a fixed order (no
        // atomics: encoder and decoder derive bit-identical gates).  The order is a property of the TILE, not of the
        // number of waves that share it (round 4, ADVICE r03: the 4- and the 8-wave build used to differ in the last bit)
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int off = 1; off < 16; off <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[r][n][e] += __shfl_xor(csum[r][n][e], off);
        float *red = lds;
        __syncthreads();  // every wave is done with the main loop's LDS
        if ((lane & 15) == 0)
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int n = 0; n < NTW; ++n) *(f32x4 *)&red[(wave * RW + r) * BN + n * 16 + (lane >> 4) * 4] = csum[r][n];
        __syncthreads();
        if (tid < BN && n0 + tid < a.Cout_pad) {
            float s = red[tid];
#pragma unroll
            for (int w = 1; w < BH; ++w) s += red[w * BN + tid];  // fixed order: tile rows 0..7
            const size_t part = (size_t)img * (a.nty * a.ntx) + (size_t)ty * a.ntx + tx;
            a.chan_partial[part * a.Cout_pad + n0 + tid] = s;
        }
    }
}

template <int KS, int NTW, int NWAVE>
int launch(K32 &k, int N, hipStream_t st) {
    constexpr int BN = 16 * NTW;
    k.ntx = (k.W + 31) / 32;
    k.nty = (k.H + 7) / 8;
    if (k.ty0 < 0 || k.ty0 >= k.nty) return DCVC_E_ARG;
    const int rows = k.band_rows > 0 ? (k.band_rows < k.nty - k.ty0 ? k.band_rows : k.nty - k.ty0) : k.nty;
    dim3 grid((unsigned)(k.ntx * (k.Cout_pad / BN)), (unsigned)rows, (unsigned)N);
    hipLaunchKernelGGL((conv_k32<KS, NTW, NWAVE>), grid, dim3(64 * NWAVE), 0, st, k);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}

int g_k32_waves = 8;

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

extern "C" int dcvc_conv_k32_set_waves(int32_t waves) {
    if (waves != 4 && waves != 8) return DCVC_E_ARG;
    g_k32_waves = waves;
    return DCVC_OK;
}

extern "C" int dcvc_conv2d_k32(const dcvc_conv_args *a, void *stream) {
    if (!a || a->nseg < 1 || a->nseg > DCVC_MAX_SEG || !a->out || !a->wpack || !a->bpack) return DCVC_E_ARG;
    if (a->stride != 1 || (a->ks != 1 && a->ks != 3) || a->precision != DCVC_PREC_FP16X3) return DCVC_E_ARG;
    if (a->Cout_pad % 32 || a->Cout > a->Cout_pad || (a->pixel_shuffle && (a->Cout & 3))) return DCVC_E_ARG;
    if (a->out_act < 0 || a->out_act > 2) return DCVC_E_ARG;  // (the mask epilogue, out_act 3, is dcvc_conv2d's)
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
    {  // output / residual images are addressed with 32-bit byte offsets (and as raw buffers)
        const unsigned long long opix = (unsigned long long)a->Hin * a->Win * (a->pixel_shuffle ? 4 : 1);
        // the kernel forms pixel index x channel stride with 24-bit multiplies (load_patch, out_off): both factors
        // must stay below 2^24 -- a 6144x3456 picture (21 M pixels) passes the 4 GiB tests below with 32..63 channels
        if (opix >= (1ull << 24) || (unsigned long long)a->out_cs * 4ull >= (1ull << 24) ||
            (a->res && (unsigned long long)a->res_cs * 4ull >= (1ull << 24)) ||
            (a->res2 && (unsigned long long)a->res2_cs * 4ull >= (1ull << 24)))
            return DCVC_E_ARG;
        for (int s = 0; s < a->nseg; ++s)
            if ((unsigned long long)a->seg[s].cs * 4ull >= (1ull << 24)) return DCVC_E_ARG;
        if (opix * a->out_cs * 4ull > 0xfffffff0ull || (a->res && opix * a->res_cs * 4ull > 0xfffffff0ull) ||
            (a->res2 && opix * a->res2_cs * 4ull > 0xfffffff0ull))
            return DCVC_E_ARG;
    }
    if (a->in_act && !(a->in_slope >= 0.f && a->in_slope <= 1.f)) return DCVC_E_ARG;  // (store_patch's max(s, s * slope))
    for (int s = 0; s < a->nseg; ++s)  // an image of a segment is addressed with 32-bit byte offsets (and as a raw buffer)
        if ((unsigned long long)a->Hin * a->Win * a->seg[s].cs * 4ull > 0xfffffff0ull) return DCVC_E_ARG;
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
    k.ty0 = a->tile_rows > 0 ? a->tile_row0 : 0;
    k.band_rows = a->tile_rows > 0 ? a->tile_rows : 0;
    const int cfin = a->pixel_shuffle ? a->Cout / 4 : a->Cout;
    // this kernel has the 16-byte epilogue only (every layer it is meant for qualifies); others stay on dcvc_conv2d
    if ((cfin % 4) || !aligned16(a->out, a->out_cs) || !aligned16(a->res, a->res_cs) || !aligned16(a->res2, a->res2_cs) ||
        (a->res_gate && (((uintptr_t)a->res_gate) & 15)))
        return DCVC_E_ARG;
    if (a->chan_partial && a->pixel_shuffle) return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = (a->Cout_pad % 64) == 0;
    // the wide 3x3 kernel runs as 8-wave workgroups (four waves per SIMD with two resident workgroups);
    // dcvc_conv_k32_set_waves(4) is the developer A/B switch back to 4 waves (bit-identical results: the sums inside a
    // tile are ordered by tile row, not by wave -- tests/test_gpu_kernels.py)
    const bool w8 = g_k32_waves != 4;
    if (a->ks == 3) return wide ? (w8 ? launch<3, 4, 8>(k, a->N, st) : launch<3, 4, 4>(k, a->N, st)) : launch<3, 2, 4>(k, a->N, st);
    return wide ? launch<1, 4, 4>(k, a->N, st) : launch<1, 2, 4>(k, a->N, st);
}
