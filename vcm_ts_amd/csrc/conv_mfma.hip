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
    int *status;  // optional: flag outputs a split-fp16 consumer would clamp
    float *chan_partial;  // optional: per-workgroup channel sums of the stored output (SE layer)
    int ntx;              // tiles per row (gridDim.x may be padded, see launch())
    int ty0, nty;         // first tile row of this launch's band, tile rows of the whole picture
    int band_rows;        // tile rows of the band (0: all)
};

constexpr float ACT_LIMIT = 65504.f / 8.f;  // F16_MAX / ACT_SCALE

__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : v * slope; }

// Developer A/B switches of this file change launch geometry only (never a result bit) and are honoured only in a process
// that declares itself a developer run with DCVC_DEV=1, like the routing switches of vcm_ts_amd/engine.py (a
// stray variable must not reconfigure a product process).
inline long dev_env_long(const char *name, long dflt) {
    const char *dev = getenv("DCVC_DEV");
    if (!dev || strcmp(dev, "1") != 0) return dflt;
    const char *v = getenv(name);
    return v ? atol(v) : dflt;
}


// PAIR (7x7, one segment of <= 8 input channels, split mode: SpyNet's first layer): a 16-deep K step would be half
// zero padding.  Instead the LDS record of patch pixel (y, x) carries the 8 channels of (y, x) in its lower half and
// the 8 channels of (y, x + 1) in its upper half, and the filter is packed in tap PAIRS (kx = 2j, 2j + 1; the pair of
// kx = 6 is a zero tap): 4 K steps per filter row instead of 7, the fragment reads unchanged.  Weights from
// dcvc_conv_pack_weights_paired; selected by dcvc_conv_args.pair_taps.
template <int KS, int S, int RPW, int NT, bool SPLIT, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void conv_mfma(const ConvK a) {
    static_assert(!PAIR || (KS == 7 && S == 1 && SPLIT), "tap pairing is built for the 7x7 split-fp16 layers");
    constexpr int BH = 4 * RPW, BW = 32, BN = 32 * NT;
    constexpr int PH = (BH - 1) * S + KS, PW = (BW - 1) * S + KS, PAD = KS / 2;
    constexpr int T = PAIR ? KS * 4 : KS * KS;           // K steps of a chunk (taps, or tap pairs)
    constexpr int TPS = PAIR ? 4 : (KS == 3 && S == 1) ? 9 : KS;  // of which staged in LDS at a time
    constexpr int NST = T / TPS;
    constexpr int EPI_LD = BN + 4;  // floats per pixel row of the epilogue's transpose tile
    // epilogue: transpose tiles of the four waves, then 4 x BN floats for the fused channel sums
    constexpr int LDS_MAIN = PH * PW * LDK + TPS * 4 * BN * 4, LDS_EPI = 4 * 32 * EPI_LD, LDS_RED = 4 * BN;
    __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN > LDS_EPI + LDS_RED ? LDS_MAIN : LDS_EPI + LDS_RED];
    float *patch = lds;
    float *wl = lds + PH * PW * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbn = a.Cout_pad / BN;
    const int nb = blockIdx.x % nbn, tx = blockIdx.x / nbn;
    if (tx >= a.ntx) return;  // padding block of launch(): keeps vertically adjacent tiles on one XCD
    const int ty = blockIdx.y + a.ty0;
    const int x0 = tx * BW, y0 = ty * BH, n0 = nb * BN, img = blockIdx.z;

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
    // Per-lane patch geometry is the same for every chunk: precompute the pixel offset and an
    // in-picture bit per staged float4; out-of-picture lanes load pixel 0 (valid memory) and are
    // zeroed when the registers are written to LDS, so the loads carry no branches.
    int poff[NP];
    unsigned inpic = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int i = tid + u * 256;
        const int p = i >> 2;
        const int py = p / PW, px = p - py * PW + (PAIR ? (i & 3) >> 1 : 0);  // (PAIR: the record's upper half is the next pixel)
        const int gy = y0 * S - PAD + py, gx = x0 * S - PAD + px;
        const bool ok = i < PH * PW * 4 && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        poff[u] = ok ? gy * a.Win + gx : 0;
        inpic |= (ok ? 1u : 0u) << u;
    }
    auto load_patch = [&](const Cursor &k) {
        const int C = a.seg_C[k.s], cs = a.seg_cs[k.s];
        const float *sp = a.seg_ptr[k.s] + (size_t)img * a.Hin * a.Win * cs;
        const int c = PAIR ? (tid & 1) * 4 : k.c0 + (tid & 3) * 4;
        const int cc = c < C ? c : 0;  // chunk tail: load channel 0, zeroed at store time
#pragma unroll
        for (int u = 0; u < NP; ++u) rp[u] = *(const f32x4 *)(sp + (size_t)poff[u] * cs + cc);
    };
    auto store_patch = [&](const Cursor &k) {
        const int C = a.seg_C[k.s];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + u * 256;
            if (i < PH * PW * 4) {
                const int c = PAIR ? (i & 1) * 4 : k.c0 + (i & 3) * 4;
                f32x4 v = ((inpic >> u) & 1u) && c < C ? rp[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
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
                    // vector form: the compiler emits packed converts (v_cvt_pk_f16_f32, v_pk_*)
                    f32x4 sv = v * ACT_SCALE;
#pragma unroll
                    for (int e = 0; e < 4; ++e) sv[e] = __builtin_amdgcn_fmed3f(sv[e], -F16_MAX, F16_MAX);
                    const f16x4 hi = __builtin_convertvector(sv, f16x4);
                    const f16x4 lo = __builtin_convertvector(sv - __builtin_convertvector(hi, f32x4), f16x4);
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
        if (cur.st == 0) store_patch(cur);
        store_w();
        __syncthreads();
        Cursor nxt = cur;
        advance(nxt);
        if (nxt.s < a.nseg) {
            if (nxt.st == 0) load_patch(nxt);
            load_w(nxt);
        }
        const int a_st = (TPS == T) ? 0 : cur.st * PW * LDK;  // staged by filter row
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int ky = (TPS == T) ? tt / KS : 0, kx = PAIR ? 2 * tt : (TPS == T) ? tt % KS : tt;
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

    // ---- epilogue: bias, activation, (gated) residual(s), NHWC or pixel-shuffled store.
    const int col = lane & 31, hh = lane >> 5;
    const int Cq = a.Cout >> 2;
    const int Cfin = a.ps ? Cq : a.Cout;
    const int Ho = a.ps ? a.Hout * 2 : a.Hout, Wo = a.ps ? a.Wout * 2 : a.Wout;
    const float inv_scale = SPLIT ? 1.f / (ACT_SCALE * WGT_SCALE) : 1.f;
    float vmax = 0.f;  // largest |output| this lane stores: the range guard (two v_max3_f32 per four outputs; an infinity is
                       // caught, a NaN can only follow one)
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
        // All residual loads of the wave's RPW rows go out first (the registers of the main loop's
        // prefetch are free now), so their latency is paid once and overlaps the LDS transposes.
        size_t pix[RPW][NIT];
        bool ok[RPW][NIT];
        f32x4 rv[RPW][NIT], rv2[RPW][NIT];
        f32x4 csum = {0.f, 0.f, 0.f, 0.f};  // this lane's 4 channels summed over its pixels (SE squeeze)
#pragma unroll
        for (int m = 0; m < RPW; ++m) {
            const int oy = y0 + wave * RPW + m;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int ox = x0 + it * PPI + pl;
                ok[m][it] = ch_ok && oy < a.Hout && ox < a.Wout;
                pix[m][it] = a.ps ? ((size_t)(img * Ho + 2 * oy + dy) * Wo + 2 * ox + dx) : ((size_t)(img * Ho + oy) * Wo + ox);
                rv[m][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                rv2[m][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok[m][it] && a.res) rv[m][it] = *(const f32x4 *)&a.res[pix[m][it] * a.res_cs + cf];
                if (ok[m][it] && a.res2) rv2[m][it] = *(const f32x4 *)&a.res2[pix[m][it] * a.res2_cs + cf];
            }
        }
#pragma unroll
        for (int m = 0; m < RPW; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    epi[((r & 3) + 8 * (r >> 2) + 4 * hh) * EPI_LD + n * 32 + col] = acc[m][n][r];
            // the transpose tile is private to this wave and a wave's LDS operations execute in
            // order: draining its own ds_writes is all the synchronisation the reads below need
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
                if (a.out_act == 3) {  // res2 is a MASK SOURCE (dcvc_hip.h): v = v * LeakyReLU'(res2) [+ res]
#pragma unroll
                    for (int e = 0; e < 4; ++e) {  // (with a residual: ONE rounding, as dcvc_mask_accumulate's fma)
                        const float mk = rv2[m][it][e] > 0.f ? 1.f : a.out_slope;
                        v[e] = a.res ? __builtin_fmaf(v[e], mk, rv[m][it][e]) : v[e] * mk;
                    }
                } else {
                    if (a.res) {
                        if (a.res_gate) {  // explicit fma: the same rounding in every kernel that applies the SE gate
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rv[m][it][e], gate[e], v[e]);
                        } else {
                            v = v + rv[m][it];
                        }
                    }
                    if (a.res2) v = rv2[m][it] + v;
                }
                if (a.chan_partial && ok[m][it]) csum += v;
                if (a.status && ok[m][it])
                    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if (ok[m][it]) *(f32x4 *)&a.out[pix[m][it] * a.out_cs + cf] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the tile is rewritten
        }
        if (a.status && !(vmax <= ACT_LIMIT)) atomicOr(a.status, DCVC_STATUS_ACT_SATURATED);
        if (a.chan_partial) {
            // SELayer's AdaptiveAvgPool (video_net.py:149-162) rides on the producing convolution: lanes with the
            // same channel quad are LPP apart -> butterfly inside the wave, the four waves through LDS, one
            // partial row per workgroup; dcvc_channel_mean_finish adds the rows in a fixed order (no atomics:
            // encoder and decoder derive bit-identical gates)
#pragma unroll
            for (int off = LPP; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], off);
            float *red = lds + LDS_EPI;
            if (lane < LPP) *(f32x4 *)&red[wave * BN + c4] = csum;
            __syncthreads();
            if (tid < BN && n0 + tid < a.Cout_pad) {
                const float s = ((red[tid] + red[BN + tid]) + red[2 * BN + tid]) + red[3 * BN + tid];
                const size_t part = (size_t)img * (a.nty * a.ntx) + (size_t)ty * a.ntx + tx;
                a.chan_partial[part * a.Cout_pad + n0 + tid] = s;
            }
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
                if (a.out_act == 3) {
                    const float mk = rv2[r] > 0.f ? 1.f : a.out_slope;
                    v = a.res ? __builtin_fmaf(v, mk, rv[r]) : v * mk;
                } else {
                    if (a.res) v = a.res_gate ? __builtin_fmaf(rv[r], gate, v) : v + rv[r];
                    if (a.res2) v = rv2[r] + v;
                }
                if (a.status) vmax = fmaxf(vmax, fabsf(v));
                a.out[pix * a.out_cs + cf] = v;
            }
        }
    }
    if (a.status && !(vmax <= ACT_LIMIT)) atomicOr(a.status, DCVC_STATUS_ACT_SATURATED);
}


template <int KS, int S, int RPW, int NT, bool PAIR = false>
int launch(ConvK &k, int N, hipStream_t st, int precision) {
    constexpr int BH = 4 * RPW, BN = 32 * NT;
    k.ntx = (k.Wout + 31) / 32;
    // Workgroups are dealt to the 8 XCDs round-robin in linear block order.  Rounding the row length up to a
    // multiple of 8 (the surplus blocks exit at once) puts tile (ty, tx) and the tile below it on the SAME XCD,
    // a row of blocks apart, i.e. resident together: the two halo rows they share are then L2 hits instead of
    // fabric reads (each XCD has its own L2).  Measured on 64->64 3x3 at 1088x1920 (tools/xcd_ab.sh): FETCH_SIZE
    // -15 % (traffic 1.32x -> 1.16x of the algorithmic bytes) but the launch takes 3 % LONGER (0.513 -> 0.528 ms):
    // the re-reads were Infinity-Cache hits already, and at the power cap time follows energy,
    // not fabric requests.  Off; DCVC_DEV=1 DCVC_XCD_PAD=1 turns it on (developer A/B).
    static const bool xcd_pad = dev_env_long("DCVC_XCD_PAD", 0) != 0;
    unsigned gx = (unsigned)(k.ntx * (k.Cout_pad / BN));
    if (xcd_pad && gx >= 8) gx = (gx + 7) & ~7u;
    k.nty = (k.Hout + BH - 1) / BH;
    if (k.ty0 < 0 || k.ty0 >= k.nty) return DCVC_E_ARG;
    const int rows = k.band_rows > 0 ? (k.band_rows < k.nty - k.ty0 ? k.band_rows : k.nty - k.ty0) : k.nty;
    dim3 grid(gx, (unsigned)rows, (unsigned)N);
    if (PAIR)
        hipLaunchKernelGGL((conv_mfma<KS, S, RPW, NT, true, PAIR>), grid, dim3(256), 0, st, k);
    else if (precision == DCVC_PREC_FP16X3)
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

// Tap-paired packing of a 7x7 layer with one input segment of <= 8 channels (split fp16 only; see PAIR above):
// K step (ky, j) holds tap (ky, 2j) in channel half 0 and tap (ky, 2j + 1) in half 1 (zero for kx = 7).
// wpack rows [(ky * 4 + j)][hi h0, hi h1, lo h0, lo h1][n][8 fp16].
extern "C" int64_t dcvc_conv_pack_size_paired(int32_t Cout, int32_t Cin, int32_t *cout_pad) {
    if (Cout <= 0 || Cin <= 0 || Cin > 8) return DCVC_E_ARG;
    const int cp = round_up(Cout, 32);
    if (cout_pad) *cout_pad = cp;
    return (int64_t)7 * 4 * 4 * cp * 4;
}

extern "C" int dcvc_conv_pack_weights_paired(const float *w, const float *b, int32_t Cout, int32_t Cin, float *wpack, float *bpack) {
    int32_t cp = 0;
    const int64_t total = dcvc_conv_pack_size_paired(Cout, Cin, &cp);
    if (total < 0 || !w || !wpack || !bpack) return DCVC_E_ARG;
    memset(wpack, 0, (size_t)total * sizeof(float));
    memset(bpack, 0, (size_t)cp * sizeof(float));
    _Float16 *base = (_Float16 *)wpack;
    bool clamped = false;
    for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx) {
            const int step = ky * 4 + (kx >> 1), h = kx & 1;
            for (int c = 0; c < Cin; ++c)
                for (int n = 0; n < Cout; ++n) {
                    float sv = w[((size_t)n * Cin + c) * 49 + ky * 7 + kx] * WGT_SCALE;
                    if (!(fabsf(sv) <= F16_MAX)) {
                        clamped = true;
                        sv = sv > 0.f ? F16_MAX : -F16_MAX;
                    }
                    const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
                    base[(((size_t)step * 4 + h) * cp + n) * 8 + c] = hi;
                    base[(((size_t)step * 4 + 2 + h) * cp + n) * 8 + c] = lo;
                }
        }
    for (int n = 0; n < Cout; ++n) bpack[n] = b ? b[n] : 0.f;
    return clamped ? DCVC_E_RANGE : DCVC_OK;
}

// rows of a workgroup's output tile per (kernel size, stride): 4 * RPW of the instantiations below
static int tile_rows(int ks, int stride) { return stride == 2 ? 4 : 8; }

extern "C" int32_t dcvc_conv_tile_rows(int32_t ks, int32_t stride) { return tile_rows(ks, stride); }

extern "C" int32_t dcvc_conv_chan_partial_parts(int32_t ks, int32_t stride, int32_t Hout, int32_t Wout) {
    if (Hout <= 0 || Wout <= 0 || (stride != 1 && stride != 2)) return DCVC_E_ARG;
    const int bh = tile_rows(ks, stride);
    return ((Wout + 31) / 32) * ((Hout + bh - 1) / bh);
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
    k.status = a->status;
    k.chan_partial = a->chan_partial;
    k.ty0 = a->tile_rows > 0 ? a->tile_row0 : 0;
    k.band_rows = a->tile_rows > 0 ? a->tile_rows : 0;
    {
        const int cfin = a->pixel_shuffle ? a->Cout / 4 : a->Cout;
        auto al = [](const void *p, int cs) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (cs & 3) == 0); };
        k.vec_epi = (cfin % 4 == 0) && al(a->out, a->out_cs) && al(a->res, a->res_cs) && al(a->res2, a->res2_cs) &&
                    (a->res_gate == nullptr || (((uintptr_t)a->res_gate) & 15) == 0);
    }
    if (a->chan_partial && (!k.vec_epi || a->pixel_shuffle)) return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = (a->Cout_pad % 64) == 0;
    const int key = a->ks * 10 + a->stride;
    // Small launches (round 4): a stride-1 layer whose 8-row tiles would not even put 1.5 workgroups on every CU runs on
    // 4-row tiles instead -- twice the workgroups, each half the work.  The 1/16- and 1/64-resolution stages of a 1080p
    // picture (36 tiles of 8 x 32 pixels) and most layers of a 256 x 256 training batch are such launches.  A tile's
    // shape changes nothing in an output's sum (chunks and taps in the same order): bit-identical, whatever the batch
    // size decides here (tests/test_gpu_kernels.py).  Not for the fused channel sums (their partial rows are per tile)
    // nor for band launches (the band unit is the 8-row tile).
    if (a->stride == 1 && !a->pair_taps && !a->chan_partial && a->tile_rows <= 0) {
        const long wgs8 = (long)((k.Wout + 31) / 32) * ((k.Hout + 7) / 8) * (a->Cout_pad / (wide ? 64 : 32)) * a->N;
        // (developer A/B of the threshold: DCVC_DEV=1 DCVC_ROWS4_WGS=n, read once; any value gives the same bits)
        static const long rows4_below = dev_env_long("DCVC_ROWS4_WGS", 384);
        if (wgs8 < rows4_below) {
            if (a->ks == 3) return wide ? launch<3, 1, 1, 2>(k, a->N, st, a->precision) : launch<3, 1, 1, 1>(k, a->N, st, a->precision);
            if (a->ks == 7) return wide ? launch<7, 1, 1, 2>(k, a->N, st, a->precision) : launch<7, 1, 1, 1>(k, a->N, st, a->precision);
            if (a->ks == 1) return wide ? launch<1, 1, 1, 2>(k, a->N, st, a->precision) : launch<1, 1, 1, 1>(k, a->N, st, a->precision);
        }
    }
    switch (key) {
        case 11: return wide ? launch<1, 1, 2, 2>(k, a->N, st, a->precision) : launch<1, 1, 2, 1>(k, a->N, st, a->precision);
        case 12: return wide ? launch<1, 2, 1, 2>(k, a->N, st, a->precision) : launch<1, 2, 1, 1>(k, a->N, st, a->precision);
        case 31:
            return wide ? launch<3, 1, 2, 2>(k, a->N, st, a->precision) : launch<3, 1, 2, 1>(k, a->N, st, a->precision);
        case 32: return wide ? launch<3, 2, 1, 2>(k, a->N, st, a->precision) : launch<3, 2, 1, 1>(k, a->N, st, a->precision);
        case 71:
            if (a->pair_taps) {  // weights from dcvc_conv_pack_weights_paired
                if (a->nseg != 1 || a->seg[0].C > 8 || a->precision != DCVC_PREC_FP16X3 || a->pixel_shuffle) return DCVC_E_ARG;
                return wide ? launch<7, 1, 2, 2, true>(k, a->N, st, a->precision) : launch<7, 1, 2, 1, true>(k, a->N, st, a->precision);
            }
            return wide ? launch<7, 1, 2, 2>(k, a->N, st, a->precision) : launch<7, 1, 2, 1>(k, a->N, st, a->precision);
        default: return DCVC_E_ARG;
    }
}
