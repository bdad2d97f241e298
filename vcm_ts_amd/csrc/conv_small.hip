// conv_small.hip -- 3x3 / 7x7 stride-1 convolutions with AT MOST 16 OUTPUT CHANNELS (split-fp16 mode).
//
// SpyNet's MEBasic ends in 32->16 and 16->2 7x7 layers and the reconstruction in a 64->3 3x3 layer
// (/root/reference/DCVC_HEM/src/models/video_net.py:99-115, video_model.py:115-128).  conv_mfma tiles 32
// output channels per MFMA, so these layers spent half / 94 % / 91 % of their matrix work on padding.
// Here the tile is 16 pixels x 16 output channels on v_mfma_f32_16x16x32_f16, and the 32-deep K of that
// instruction carries the operand SPLIT instead of more channels:
//     A (pixels)  = [xh ch0-7 | xh ch8-15 | xl ch0-7 | xl ch8-15]      one 16-byte slot per lane group
//     B1 (filter) = [wh ch0-7 | wh ch8-15 | wh ch0-7 | wh ch8-15]  ->  xh*wh + xl*wh in ONE instruction
//     B2 (filter) = [wl ch0-7 | wl ch8-15 |    0     |     0     ]  ->  xh*wl           (same A fragment)
// i.e. the three products of DCVC_PREC_FP16X3 (conv_mfma.hip) in two 16-cycle MFMAs per tap and 16-channel
// chunk, one ds_read_b128 per pixel tile.  Same operand values, fp32 accumulation; the accumulation order
// differs from conv_mfma's (results agree to fp32 rounding, encoder and decoder both take this kernel).
//
// Structure as conv_mfma: a 256-thread workgroup owns 8 rows x 32 pixels, fp32 NHWC inputs are converted to
// (hi, lo) while they are staged (register prefetch of the next stage behind the MFMAs), filter slab per tap
// row.  LDS: 64-byte pixel records whose four slots are XOR-swizzled with (pixel>>1)&3 -- conflict-free for
// the 16-pixel x 4-slot fragment read at every tap offset -- 34 KB + 7 KB for 7x7: three workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "dcvc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 16;
constexpr float ACT_SCALE = 8.f, WGT_SCALE = 64.f, F16_MAX = 65504.f;
constexpr float ACT_LIMIT = F16_MAX / ACT_SCALE;

struct SmallK {
    const float *seg_ptr[DCVC_MAX_SEG];
    int seg_C[DCVC_MAX_SEG];
    int seg_cs[DCVC_MAX_SEG];
    int nseg;
    int H, W;
    int in_act;
    float in_slope;
    const char *wpack;
    const float *bpack;
    int Cout;
    float *out;
    int out_cs, out_act;
    float out_slope;
    const float *res;
    int res_cs;
    int *status;
};

__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int KS>
__global__ __launch_bounds__(256, 3) void conv_small(const SmallK a) {
    constexpr int RPW = 2, BH = 4 * RPW, BW = 32, PAD = KS / 2, PH = BH + KS - 1, PW = BW + KS - 1, T = KS * KS;
    constexpr int TPS = KS, NST = KS;               // one filter row per stage
    constexpr int PATCH_F = PH * PW * 16;           // floats: 64-byte records
    constexpr int SLAB_F = TPS * 4 * 16 * 4;        // floats: [tap][slot][16 cout][16 B]
    __shared__ __attribute__((aligned(16))) float lds[PATCH_F + SLAB_F];
    char *patch = (char *)lds;
    float *wl = lds + PATCH_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH, img = blockIdx.z;
    const int pr = lane & 15, g = lane >> 4;

    f32x4 acc[RPW][2];
#pragma unroll
    for (int m = 0; m < RPW; ++m)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) acc[m][hf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NP = (PH * PW * 4 + 255) / 256, NW = (SLAB_F / 4 + 255) / 256;
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
        const int p = i >> 2;
        const int py = p / PW, px = p - py * PW;
        const int gy = y0 - PAD + py, gx = x0 - PAD + px;
        const bool ok = i < PH * PW * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        poff[u] = ok ? gy * a.W + gx : 0;
        inpic |= (ok ? 1u : 0u) << u;
    }
    auto load_patch = [&](const Cursor &k) {
        const int C = a.seg_C[k.s], cs = a.seg_cs[k.s];
        const float *sp = a.seg_ptr[k.s] + (size_t)img * a.H * a.W * cs;
        const int c = k.c0 + (tid & 3) * 4;
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
                const int qd = i & 3, p = i >> 2;
                const int c = k.c0 + qd * 4;
                f32x4 v = ((inpic >> u) & 1u) && c < C ? rp[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (c + 3 >= C) {
                    if (c + 1 >= C) v[1] = 0.f;
                    if (c + 2 >= C) v[2] = 0.f;
                    v[3] = 0.f;
                }
                if (a.in_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act(v[e], a.in_slope);
                }
                f32x4 sv = v * ACT_SCALE;
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] = __builtin_amdgcn_fmed3f(sv[e], -F16_MAX, F16_MAX);
                const f16x4 hi = __builtin_convertvector(sv, f16x4);
                const f16x4 lo = __builtin_convertvector(sv - __builtin_convertvector(hi, f32x4), f16x4);
                // record = 4 slots [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15], slot index XOR (pixel >> 1) & 3
                const int sw = (p >> 1) & 3, h = qd >> 1;
                char *rec = patch + p * 64 + (qd & 1) * 8;
                *(f16x4 *)(rec + ((h ^ sw) << 4)) = hi;
                *(f16x4 *)(rec + (((2 + h) ^ sw) << 4)) = lo;
            }
        }
    };
    auto load_w = [&](const Cursor &k) {
        const f32x4 *wsrc = (const f32x4 *)(a.wpack + ((size_t)(k.cg * T + k.st * TPS)) * 1024);
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * 256;
            if (i < SLAB_F / 4) rw[u] = wsrc[i];
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * 256;
            if (i < SLAB_F / 4) *(f32x4 *)&wl[i * 4] = rw[u];
        }
    };

    Cursor cur = {0, 0, 0, 0};
    load_patch(cur);
    load_w(cur);
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    while (cur.s < a.nseg) {
        __syncthreads();  // every wave is done reading the previous stage's LDS
        if (cur.st == 0) store_patch(cur);
        store_w();
        __syncthreads();
        Cursor nxt = cur;
        advance(nxt);
        if (nxt.s < a.nseg) {
            if (nxt.st == 0) load_patch(nxt);
            load_w(nxt);
        }
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            // filter fragments of this tap: B1 = wh of channel half g&1 (for all four lane groups), B2 = wl (g < 2)
            const f16x8 b1 = *(const f16x8 *)&wl[((tt * 4 + (g & 1)) * 16 + pr) * 4];
            const f16x8 b2 = g < 2 ? *(const f16x8 *)&wl[((tt * 4 + 2 + g) * 16 + pr) * 4] : zero8;
#pragma unroll
            for (int m = 0; m < RPW; ++m)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int p = (wave * RPW + m + cur.st) * PW + hf * 16 + pr + tt;
                    const f16x8 av = *(const f16x8 *)(patch + p * 64 + ((g ^ ((p >> 1) & 3)) << 4));
                    acc[m][hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b1, acc[m][hf], 0, 0, 0);
                    acc[m][hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b2, acc[m][hf], 0, 0, 0);
                }
        }
        cur = nxt;
    }

    // ---- epilogue: lane (pr = channel, g): pixels 4g .. 4g+3 of each 16-pixel tile
    const float inv_scale = 1.f / (ACT_SCALE * WGT_SCALE);
    const int ch = pr;
    bool sat = false;
    if (ch < a.Cout) {
        const float bias = a.bpack[ch];
#pragma unroll
        for (int m = 0; m < RPW; ++m) {
            const int oy = y0 + wave * RPW + m;
            if (oy >= a.H) continue;
            float rv[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // residual loads first (res may alias out)
                    const int ox = x0 + hf * 16 + g * 4 + j;
                    rv[hf][j] = (a.res && ox < a.W) ? a.res[((size_t)(img * a.H + oy) * a.W + ox) * a.res_cs + ch] : 0.f;
                }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ox = x0 + hf * 16 + g * 4 + j;
                    if (ox >= a.W) continue;
                    float v = acc[m][hf][j] * inv_scale + bias;
                    if (a.out_act == 1) v = act(v, a.out_slope);
                    else if (a.out_act == 2) v = fminf(fmaxf(v, 0.f), 1.f);
                    if (a.res) v += rv[hf][j];
                    if (a.status && !(fabsf(v) <= ACT_LIMIT)) sat = true;
                    a.out[((size_t)(img * a.H + oy) * a.W + ox) * a.out_cs + ch] = v;
                }
        }
    }
    if (sat) atomicOr(a.status, DCVC_STATUS_ACT_SATURATED);
}

}  // namespace

extern "C" int64_t dcvc_conv_small_pack_bytes(int32_t Cout, int32_t ks, int32_t nseg, const int32_t *seg_C) {
    if (Cout <= 0 || Cout > 16 || nseg <= 0 || nseg > DCVC_MAX_SEG || (ks != 3 && ks != 7)) return DCVC_E_ARG;
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) {
        if (seg_C[s] <= 0) return DCVC_E_ARG;
        chunks += (seg_C[s] + KC - 1) / KC;
    }
    return (int64_t)chunks * ks * ks * 1024;
}

// wpack as fp16: [chunk][tap][slot: hi k0-7, hi k8-15, lo k0-7, lo k8-15][16 output channels][8]; bpack: 16 floats
extern "C" int dcvc_conv_small_pack_weights(const float *w, const float *b, int32_t Cout, int32_t ks, int32_t nseg,
                                            const int32_t *seg_C, void *wpack, float *bpack) {
    const int64_t total = dcvc_conv_small_pack_bytes(Cout, ks, nseg, seg_C);
    if (total < 0 || !w || !wpack || !bpack) return DCVC_E_ARG;
    const int T = ks * ks;
    int Cin = 0;
    for (int s = 0; s < nseg; ++s) Cin += seg_C[s];
    memset(wpack, 0, (size_t)total);
    for (int n = 0; n < 16; ++n) bpack[n] = (b && n < Cout) ? b[n] : 0.f;
    _Float16 *base = (_Float16 *)wpack;
    int status = DCVC_OK, cg = 0, cin0 = 0;
    for (int s = 0; s < nseg; ++s) {
        for (int c0 = 0; c0 < seg_C[s]; c0 += KC, ++cg)
            for (int t = 0; t < T; ++t)
                for (int cc = 0; cc < KC && c0 + cc < seg_C[s]; ++cc)
                    for (int n = 0; n < Cout; ++n) {
                        float sv = w[((size_t)n * Cin + cin0 + c0 + cc) * T + t] * WGT_SCALE;
                        if (!(sv <= F16_MAX && sv >= -F16_MAX)) {
                            status = DCVC_E_RANGE;
                            sv = sv > 0 ? F16_MAX : -F16_MAX;
                        }
                        const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
                        const size_t blk = ((size_t)cg * T + t) * 4;
                        base[((blk + (cc >> 3)) * 16 + n) * 8 + (cc & 7)] = hi;
                        base[((blk + 2 + (cc >> 3)) * 16 + n) * 8 + (cc & 7)] = lo;
                    }
        cin0 += seg_C[s];
    }
    return status;
}

// dcvc_conv_args as for dcvc_conv2d, restricted to: Cout <= 16, ks 3 or 7, stride 1, no pixel shuffle, no gate,
// no second residual, no chan_partial; wpack / bpack from dcvc_conv_small_pack_weights; DCVC_PREC_FP16X3.
extern "C" int dcvc_conv2d_small(const dcvc_conv_args *a, void *stream) {
    if (a && (a->out_act < 0 || a->out_act > 2)) return DCVC_E_ARG;  // (the mask epilogue, out_act 3, is dcvc_conv2d's)
    if (!a || a->nseg < 1 || a->nseg > DCVC_MAX_SEG || !a->out || !a->wpack || !a->bpack) return DCVC_E_ARG;
    if (a->tile_rows > 0) return DCVC_E_ARG;  // no banded launches for this kernel
    if (a->Cout <= 0 || a->Cout > 16 || (a->ks != 3 && a->ks != 7) || a->stride != 1 || a->pixel_shuffle || a->res_gate ||
        a->res2 || a->chan_partial || a->precision != DCVC_PREC_FP16X3 || a->N <= 0)
        return DCVC_E_ARG;
    SmallK k;
    memset(&k, 0, sizeof(k));
    for (int s = 0; s < a->nseg; ++s) {
        if (!a->seg[s].ptr || (a->seg[s].cs & 3) || a->seg[s].cs < ((a->seg[s].C + 3) & ~3) || ((uintptr_t)a->seg[s].ptr & 15))
            return DCVC_E_ARG;
        k.seg_ptr[s] = a->seg[s].ptr;
        k.seg_C[s] = a->seg[s].C;
        k.seg_cs[s] = a->seg[s].cs;
    }
    k.nseg = a->nseg;
    k.H = a->Hin, k.W = a->Win;
    k.in_act = a->in_act, k.in_slope = a->in_slope;
    k.wpack = (const char *)a->wpack, k.bpack = a->bpack;
    k.Cout = a->Cout;
    k.out = a->out, k.out_cs = a->out_cs, k.out_act = a->out_act, k.out_slope = a->out_slope;
    k.res = a->res, k.res_cs = a->res_cs;
    k.status = a->status;
    dim3 grid((unsigned)((a->Win + 31) / 32), (unsigned)((a->Hin + 7) / 8), (unsigned)a->N);
    if (a->ks == 7)
        hipLaunchKernelGGL(conv_small<7>, grid, dim3(256), 0, (hipStream_t)stream, k);
    else
        hipLaunchKernelGGL(conv_small<3>, grid, dim3(256), 0, (hipStream_t)stream, k);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}
