// conv_k32ws.hip -- EXPERIMENT (round 3): conv_k32's 3x3 / 64-channel-block kernel as ONE persistent, wave-specialised
// workgroup per CU.  8 consumer waves only read fragments and issue MFMAs (+ the epilogue of their tile row); 4 producer
// waves only move data: filter rows and the next chunk's patch go global -> registers -> (split fp16) -> LDS into the
// OTHER buffer of a double-buffered pair while the consumers work on the current one.  One s_barrier per step, at a
// single site both roles reach.  Same arithmetic, same order of products as conv_k32 (bit-identical results).
//
// Built only by tools/build_k32ws.sh into tools/probes/variants/k32ws.so, where it takes over the symbol dcvc_conv2d_k32
// (the product's entry point is renamed dcvc_conv2d_k32_base in that build) for the launches it covers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dcvc_hip.h"

extern "C" int dcvc_conv2d_k32_base(const dcvc_conv_args *a, void *stream);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef WS_STAMP
#define WS_STAMP(i)
#endif
// ablation builds: 1 = consumers skip fragment reads + MFMAs, 2 = producers only keep the barriers, 4 = no epilogue
#ifndef WS_ABLATE
#define WS_ABLATE 0
#endif
#define WS_BARRIER() do { if (!(WS_ABLATE & 16)) __syncthreads(); } while (0)

namespace {

constexpr int KC = 32, REC = 40;
constexpr float ACT_SCALE = 8.f, WGT_SCALE = 64.f, F16_MAX = 65504.f, ACT_LIMIT = F16_MAX / ACT_SCALE;
#ifndef WS_NCW
#define WS_NCW 8
#endif
constexpr int NCW = WS_NCW, NPW = 4, NTHR = 64 * (NCW + NPW), NPT = 64 * NPW;
constexpr int BH = 8, BW = 32, BN = 64, NTW = 4, RWC = BH / NCW, MT = 2 * RWC;  // a consumer wave owns RWC tile rows
constexpr int PH = BH + 2, PW = BW + 2, NPIX = PH * PW;
constexpr int PATCH_F = NPIX * REC;          // floats per patch buffer (54 400 B)
constexpr int FROW_F = 3 * 8 * BN * 4;       // floats per filter-row buffer (24 576 B)
constexpr int NQ = NPIX * 8;                 // float4s of a patch chunk
constexpr int NPS = 4;                       // patch float4 slots per producer lane per third (3 * 4 * 256 >= 2720)
constexpr int NWS = FROW_F / 4 / NPT;        // filter float4s per producer lane per row (6)
static_assert(3 * NPS * NPT >= NQ && NWS * NPT * 4 == FROW_F, "staging slots");

struct WS {
    const float *seg_ptr[DCVC_MAX_SEG];
    int seg_C[DCVC_MAX_SEG];
    int seg_cs[DCVC_MAX_SEG];
    int nseg, nchunks;
    int H, W;
    int in_act;
    float in_slope;
    const float *wpack;
    const float *bpack;
    int Cout, Cout_pad;
    float *out;
    int out_cs, out_act;
    float out_slope;
    int ps;
    const float *res;
    int res_cs;
    const float *res_gate;
    const float *res2;
    int res2_cs;
    int *status;
    int ntx, nty, nbn, nitems;  // items = N * nty * ntx * nbn
    int stagger;                // start delay per phase (blockIdx & 7), in units of 1024 cycles
};

__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : v * slope; }

// position in this workgroup's sequence of 32-channel chunks: item j of the workgroup (tile x output block), chunk cw
struct Cur {
    int j, cw, s, c0;       // item number, chunk in item, segment, first channel of the chunk in the segment
    int x0, y0, img, n0;    // decoded item
    bool live;
};

__global__ __launch_bounds__(NTHR, NTHR / 256) void conv_k32ws(const WS a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * PATCH_F + 2 * FROW_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= NCW;

    auto decode = [&](Cur &k) {
        const int item = blockIdx.x + k.j * gridDim.x;
        k.live = item < a.nitems;
        const int it = k.live ? item : 0;
        const int nb = it % a.nbn, t = it / a.nbn;
        const int tx = t % a.ntx, t2 = t / a.ntx;
        const int ty = t2 % a.nty;
        k.img = t2 / a.nty;
        k.x0 = tx * BW;
        k.y0 = ty * BH;
        k.n0 = nb * BN;
    };
    auto start = [&](Cur &k) {
        k.j = 0;
        k.cw = 0;
        k.s = 0;
        k.c0 = 0;
        decode(k);
    };
    auto advance = [&](Cur &k) {
        ++k.cw;
        k.c0 += KC;
        if (k.c0 >= a.seg_C[k.s]) {
            ++k.s;
            k.c0 = 0;
        }
        if (k.cw == a.nchunks) {
            k.cw = 0;
            k.s = 0;
            k.c0 = 0;
            ++k.j;
            decode(k);
        }
    };
    // number of items of this workgroup (the same in every wave: the loop trip count and so the barrier count)
    // de-phase the workgroups of the chip: a persistent grid started together otherwise stays in lock-step, and every CU
    // writes its tile (and fetches its patch) in the same microsecond
    for (int i = (int)(blockIdx.x & 7) * a.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(16);
    const int my_items = ((int)blockIdx.x < a.nitems) ? (a.nitems - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nci = my_items * a.nchunks;

    if (producer) {
        // ================================================================ producer waves
        const int ptid = tid - 64 * NCW;
        const unsigned lane_off = (ptid & 7) * 16u;
        int pyx[3 * NPS];  // patch pixel (py << 8 | px) of each of this lane's float4 slots, -1 beyond the patch
#pragma unroll
        for (int v = 0; v < 3 * NPS; ++v) {
            const int i = v * NPT + ptid, p = i >> 3;
            pyx[v] = i < NQ ? ((p / PW) << 8) | (p % PW) : -1;
        }
        unsigned wofs[NWS];
#pragma unroll
        for (int u = 0; u < NWS; ++u) {
            const int i = ptid + u * NPT;
            wofs[u] = (unsigned)((i / BN) * a.Cout_pad + (i % BN)) * 16u;
        }
        f32x4 rp[3][NPS], rw[3][NWS];
        unsigned okm[3] = {0, 0, 0};
        auto ld16 = [](const void *base, unsigned byte_off) __attribute__((always_inline)) {
            return *(const f32x4 *)((const char *)base + byte_off);
        };
        auto load_patch_third = [&](const Cur &k, int r) __attribute__((always_inline)) {
            const unsigned cs4 = (unsigned)a.seg_cs[k.s] * 4u;
            const char *sp = (const char *)(a.seg_ptr[k.s] + (size_t)k.img * a.H * a.W * a.seg_cs[k.s] + k.c0);
            unsigned m = 0;
#pragma unroll
            for (int u = 0; u < NPS; ++u) {
                const int c = pyx[r * NPS + u];
                const int gy = k.y0 - 1 + (c >> 8), gx = k.x0 - 1 + (c & 255);
                const bool ok = c >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const unsigned off = ok ? (unsigned)(gy * a.W + gx) : 0u;
                rp[r][u] = ld16(sp, __umul24(off, cs4) + lane_off);
                m |= (ok ? 1u : 0u) << u;
            }
            okm[r] = m;
        };
        auto store_patch_third = [&](float *patch, int r) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < NPS; ++u) {
                const int i = (r * NPS + u) * NPT + ptid;
                if (i < NQ) {
                    f32x4 v = ((okm[r] >> u) & 1u) ? rp[r][u] : (f32x4){0.f, 0.f, 0.f, 0.f};
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
        auto load_w_row = [&](const Cur &k, int r) __attribute__((always_inline)) {
            const int cg = k.cw;  // chunk index over the concatenated input == packed chunk index
            const char *wsrc = (const char *)(a.wpack + ((size_t)(cg * 9 + r * 3) * 8) * a.Cout_pad * 4 + (size_t)k.n0 * 4);
#pragma unroll
            for (int u = 0; u < NWS; ++u) rw[r][u] = ld16(wsrc, wofs[u]);
        };
        auto store_w_row = [&](float *fb, int r) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < NWS; ++u) *(f32x4 *)&fb[(ptid + u * NPT) * 4] = rw[r][u];
        };

        Cur nxt, nn;  // chunk ci + 1, chunk ci + 2
        start(nxt);
        if (nci > 0) {
            // prologue: chunk 0 entirely, rows 1 and 2 of its filter left pending in registers, chunk 1's patch requested
#pragma unroll
            for (int r = 0; r < 3; ++r) load_patch_third(nxt, r);
#pragma unroll
            for (int r = 0; r < 3; ++r) load_w_row(nxt, r);
#pragma unroll
            for (int r = 0; r < 3; ++r) store_patch_third(lds, r);
            store_w_row(lds + 2 * PATCH_F, 0);
        }
        nn = nxt;
        if (nci > 1) {
            advance(nxt);
            nn = nxt;
#pragma unroll
            for (int r = 0; r < 3; ++r) load_patch_third(nxt, r);
        }
        __syncthreads();
        for (int ci = 0; ci < nci; ++ci) {
            const bool has1 = ci + 1 < nci, has2 = ci + 2 < nci;
            if (has2) advance(nn);
            float *patch_next = lds + ((ci + 1) & 1) * PATCH_F;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int g = 3 * ci + r;
                // filter row of step g + 1 (row r + 1 of this chunk, or row 0 of the next)
                if (WS_ABLATE & 2) {
                    WS_BARRIER();
                    continue;
                }
                if (r < 2 || has1) store_w_row(lds + 2 * PATCH_F + ((g + 1) & 1) * FROW_F, (r + 1) % 3);
                if (has1) {
                    store_patch_third(patch_next, r);
                    load_w_row(nxt, r);
                }
                if (has2) load_patch_third(nn, r);
                WS_BARRIER();
            }
            nxt = nn;
        }
    } else {
        // ================================================================ consumer waves
        f32x4 acc[MT][NTW], rv[MT][NTW];
        const int a_base = (wave * RWC * PW + (lane & 15)) * REC + (lane >> 4) * 4;
        const int b_base = ((lane >> 4) * BN + (lane & 15)) * 4;
        const int Cq = a.Cout >> 2;
        const int Cfin = a.ps ? Cq : a.Cout;
        const int Ho = a.ps ? a.H * 2 : a.H, Wo = a.ps ? a.W * 2 : a.W;
        unsigned okq = 0, pixo[MT], chq[NTW], pso[NTW];
        const char *res_b = nullptr, *res2_b = nullptr;
        char *out_b = nullptr;
        auto out_off = [&](int m, int n, int cs) __attribute__((always_inline)) -> unsigned {
            return (__umul24(pixo[m] + pso[n], (unsigned)cs) + chq[n]) * 4u;
        };
        Cur cur;
        start(cur);
        __syncthreads();
        for (int ci = 0; ci < nci; ++ci) {
            const float *patch = lds + (ci & 1) * PATCH_F;
            if (cur.cw == 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // output addressing of this tile (kept for the residual prefetch and the epilogue)
                okq = 0;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int oy = cur.y0 + wave * RWC + (m >> 1);
                    const int ox = cur.x0 + (m & 1) * 16 + (lane & 15);
#pragma unroll
                    for (int n = 0; n < NTW; ++n)
                        if (oy < a.H && ox < a.W && cur.n0 + n * 16 + (lane >> 4) * 4 < a.Cout) okq |= 1u << (m * NTW + n);
                    pixo[m] = a.ps ? (unsigned)((2 * oy) * Wo + 2 * ox) : (unsigned)(oy * a.W + ox);
                }
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const int ch = cur.n0 + n * 16 + (lane >> 4) * 4;
                    const int sub = a.ps ? ch / Cq : 0;
                    chq[n] = (unsigned)(ch - sub * Cq);
                    pso[n] = (unsigned)((sub >> 1) * Wo + (sub & 1));
                }
                const size_t img_pix = (size_t)cur.img * Ho * Wo;
                res_b = a.res ? (const char *)(a.res + img_pix * a.res_cs) : nullptr;
                res2_b = a.res2 ? (const char *)(a.res2 + img_pix * a.res2_cs) : nullptr;
                out_b = (char *)(a.out + img_pix * a.out_cs);
            }
            if (a.res && cur.cw == a.nchunks - 1 && cur.live) {  // the residual is requested a whole chunk before its use
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTW; ++n)
                        rv[m][n] = *(const f32x4 *)(res_b + (((okq >> (m * NTW + n)) & 1u) ? out_off(m, n, a.res_cs) : 0u));
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int g = 3 * ci + r;
                const float *wl = lds + 2 * PATCH_F + (g & 1) * FROW_F;
#pragma unroll
                for (int tl = 0; tl < ((WS_ABLATE & 1) ? 0 : 3); ++tl) {
                    f16x8 ah[MT], al[MT], bh[NTW], bl[NTW];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float *rec = &patch[a_base + ((r + (m >> 1)) * PW + (m & 1) * 16 + tl) * REC];
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
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], ah[m], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[n], ah[m], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], al[m], acc[m][n], 0, 0, 0);
                        }
                }
                if (!(WS_ABLATE & 4) && r == 2 && cur.cw == a.nchunks - 1 && cur.live) {
                    // ---- epilogue of this wave's tile rows, straight from the accumulators (as conv_k32)
                    constexpr float inv_scale = 1.f / (ACT_SCALE * WGT_SCALE);
                    float vmax = 0.f;
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        const int ch = cur.n0 + n * 16 + (lane >> 4) * 4;
                        const f32x4 bias = *(const f32x4 *)&a.bpack[ch];
                        f32x4 gate = {1.f, 1.f, 1.f, 1.f};
                        if (a.res_gate && ch < a.Cout) gate = *(const f32x4 *)&a.res_gate[(size_t)cur.img * Cfin + (a.ps ? ch % Cq : ch)];
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
                                if (a.res_gate) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rv[m][n][e], gate[e], v[e]);
                                } else {
                                    v = v + rv[m][n];
                                }
                            }
                            if ((okq >> (m * NTW + n)) & 1u) {
                                if (a.res2) v = *(const f32x4 *)(res2_b + out_off(m, n, a.res2_cs)) + v;
                                vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                                if (!(WS_ABLATE & 8)) *(f32x4 *)(out_b + out_off(m, n, a.out_cs)) = v;
                                else asm volatile("" ::"v"(v));
                            }
                        }
                    }
                    if (a.status && !(vmax <= ACT_LIMIT)) atomicOr(a.status, DCVC_STATUS_ACT_SATURATED);
                }
                WS_BARRIER();
            }
            advance(cur);
        }
    }
}

inline bool aligned16(const void *p, int cs) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (cs & 3) == 0); }

}  // namespace

extern "C" int dcvc_conv2d_k32(const dcvc_conv_args *a, void *stream) {
    const char *env = getenv("DCVC_K32_WS");
    const bool on = !(env && atoi(env) == 0);
    if (!on || !a || a->ks != 3 || a->stride != 1 || a->precision != DCVC_PREC_FP16X3 || (a->Cout_pad % 64) || a->chan_partial ||
        a->tile_rows > 0 || a->nseg < 1 || a->nseg > DCVC_MAX_SEG)
        return dcvc_conv2d_k32_base(a, stream);
    const int cfin = a->pixel_shuffle ? a->Cout / 4 : a->Cout;
    if (!a->out || !a->wpack || !a->bpack || a->Cout > a->Cout_pad || (a->pixel_shuffle && (a->Cout & 3)) || (cfin % 4) ||
        !aligned16(a->out, a->out_cs) || !aligned16(a->res, a->res_cs) || !aligned16(a->res2, a->res2_cs) ||
        (a->res_gate && (((uintptr_t)a->res_gate) & 15)))
        return dcvc_conv2d_k32_base(a, stream);
    WS k;
    memset(&k, 0, sizeof(k));
    for (int s = 0; s < a->nseg; ++s) {
        if (!a->seg[s].ptr || a->seg[s].C <= 0 || a->seg[s].C % KC || (a->seg[s].cs & 3) || a->seg[s].cs < a->seg[s].C ||
            ((uintptr_t)a->seg[s].ptr & 15))
            return DCVC_E_ARG;
        k.seg_ptr[s] = a->seg[s].ptr;
        k.seg_C[s] = a->seg[s].C;
        k.seg_cs[s] = a->seg[s].cs;
        k.nchunks += a->seg[s].C / KC;
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
    k.ntx = (k.W + BW - 1) / BW;
    k.nty = (k.H + BH - 1) / BH;
    k.nbn = k.Cout_pad / BN;
    const long long items = (long long)a->N * k.nty * k.ntx * k.nbn;
    if (items <= 0 || items > 0x7fffffff) return DCVC_E_ARG;
    k.nitems = (int)items;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return DCVC_E_LAUNCH;
        ncu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const char *senv = getenv("DCVC_K32_WS_STAG");
    k.stagger = senv ? atoi(senv) : 0;
    const char *genv = getenv("DCVC_K32_WS_GRID");
    int grid = genv ? atoi(genv) : ncu;
    if (grid <= 0) grid = ncu;
    if (grid > k.nitems) grid = k.nitems;
    hipLaunchKernelGGL(conv_k32ws, dim3((unsigned)grid), dim3(NTHR), 0, (hipStream_t)stream, k);
    return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH;
}
