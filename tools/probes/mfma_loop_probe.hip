// Developer probe (NOT part of libdcvc_hip.so): the bare inner loops a split-fp16 3x3 convolution can be built
// from, with NO global memory traffic inside the loop -- LDS fragment reads + MFMAs only, random operands -- so that
// what the arithmetic itself costs on this board (cycles, clock held, wall time) is known before a kernel is built
// around it.  Every variant issues the same number of matrix FLOPs per launch (a 64->64 3x3 layer's three split
// products, ~10 1080p layers' worth) on tiles of the real kernels' geometry:
//
//   V0  v_mfma_f32_32x32x16_f16, A and B fragments from LDS, 64 px x 64 co per wave (RPW 2 x NT 2), 64 KB LDS,
//       2 workgroups per CU: the main loop of conv_mfma<3,1,2,2,true> as it is
//   V1  32x32x16, B (the filter) held in REGISTERS for the whole launch (288 VGPRs per wave: 32 output channels x
//       64 input channels x 9 taps, hi + lo), A from LDS, 128 px x 32 co per wave, 1 workgroup per CU, 512 VGPRs
//   V2  v_mfma_f32_16x16x32_f16 with 32-channel chunks (one tap = one K step), A and B from LDS, the same
//       64 px x 64 co per wave as V0, 2 workgroups per CU          
//   V3  16x16x32, B in registers (288 VGPRs), A from LDS, 128 px x 32 co per wave, 1 workgroup per CU
//
// Reported per variant: median / min wall time per launch over interleaved rounds in one process, matrix TFLOP/s
// against the 2500 TFLOP/s datasheet figure, and the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz,
// median over workgroups) -- MI355X_MICROARCH.md "DVFS give-back" items 6 and 7.
// Build: make -C tools/probes mfma_loop_probe ; run on the GPU box: tools/probes/mfma_loop_probe [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// two random fp16 values in (-2, 2) with random mantissas, as one dword
__device__ __forceinline__ uint32_t rnd_pair(uint32_t i, uint32_t seed) {
    const uint32_t h = mix(i * 2654435761U + seed);
    const float a = ((int)(h & 0xffff) - 32768) * (1.f / 16384.f), b = ((int)(h >> 16) - 32768) * (1.f / 16384.f);
    union { _Float16 f[2]; uint32_t u; } v;
    v.f[0] = (_Float16)a; v.f[1] = (_Float16)b;
    return v.u;
}
__device__ __forceinline__ void fill_lds(uint32_t *lds, int ndw, uint32_t seed) {
    for (int i = threadIdx.x; i < ndw; i += blockDim.x) lds[i] = rnd_pair(i + blockIdx.x * 7919u, seed);
    __syncthreads();
}
struct Stamp { unsigned long long cyc, rt; };
#define STAMP_BEGIN() const unsigned long long c0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime()
#define STAMP_END(st)                                                                                     \
    if (threadIdx.x == 0) {                                                                               \
        st[blockIdx.x].cyc = __builtin_amdgcn_s_memtime() - c0_;                                          \
        st[blockIdx.x].rt = __builtin_amdgcn_s_memrealtime() - r0_;                                       \
    }

// ---------------------------------------------------------------- V0: conv_mfma<3,1,2,2,true>'s loop
__global__ __launch_bounds__(256, 2) void v0_kernel(float *sink, Stamp *st, int iters, uint32_t seed) {
    constexpr int PW = 34, LDK = 20, BN = 64;
    __shared__ __attribute__((aligned(16))) float lds[10 * PW * LDK + 9 * 4 * BN * 4];
    fill_lds((uint32_t *)lds, sizeof(lds) / 4, seed);
    const float *patch = lds, *wl = lds + 10 * PW * LDK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int a_base = ((wave * 2) * PW + (lane & 31)) * LDK + (lane >> 5) * 4;
    const int b_base = ((lane >> 5) * BN + (lane & 31)) * 4;
    STAMP_BEGIN();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) {
            const int ky = tt / 3, kx = tt % 3;
            f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float *rec = &patch[a_base + ((m + ky) * PW + kx) * LDK];
                ah[m] = *(const f16x8 *)rec;
                al[m] = *(const f16x8 *)(rec + 8);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[n] = *(const f16x8 *)&wl[b_base + ((tt * 4) * BN + n * 32) * 4];
                bl[n] = *(const f16x8 *)&wl[b_base + ((tt * 4 + 2) * BN + n * 32) * 4];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
    }
    STAMP_END(st);
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------- V1: filter in registers, 32x32x16
__global__ __launch_bounds__(256, 1) void v1_kernel(const f16x8 *wreg, float *sink, Stamp *st, int iters, uint32_t seed) {
    constexpr int PW = 34, LDK = 20, CH = 10 * PW * LDK;  // one 16-channel patch buffer, floats
    __shared__ __attribute__((aligned(16))) float lds[4 * CH];
    fill_lds((uint32_t *)lds, sizeof(lds) / 4, seed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = wave & 1, half = wave >> 1;
    f16x8 bh[4][9], bl[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            bh[c][t] = wreg[(((c * 9 + t) * 2 + nt) * 2 + 0) * 64 + lane];
            bl[c][t] = wreg[(((c * 9 + t) * 2 + nt) * 2 + 1) * 64 + lane];
        }
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const int a_base = ((half * 4) * PW + (lane & 31)) * LDK + (lane >> 5) * 4;
    STAMP_BEGIN();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) {
                const int ky = tt / 3, kx = tt % 3;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float *rec = &lds[c * CH + a_base + ((m + ky) * PW + kx) * LDK];
                    const f16x8 ah = *(const f16x8 *)rec, al = *(const f16x8 *)(rec + 8);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[c][tt], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[c][tt], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[c][tt], acc[m], 0, 0, 0);
                }
            }
    }
    STAMP_END(st);
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------- V2: 16x16x32, 32-channel chunks, A and B from LDS
__global__ __launch_bounds__(256, 2) void v2_kernel(float *sink, Stamp *st, int iters, uint32_t seed) {
    // patch record of a pixel: 32 channels = [hi 64 B | lo 64 B | 32 B pad] (160-B stride: conflict-free b128 reads);
    // filter slab of one filter row (3 taps): [tap][hi, lo][kq 4][64 co][16 B]
    constexpr int PW = 34, REC = 40, PATCH = 10 * PW * REC, WROW = 3 * 2 * 4 * 64 * 4;
    __shared__ __attribute__((aligned(16))) float lds[PATCH + WROW];
    fill_lds((uint32_t *)lds, sizeof(lds) / 4, seed);
    const float *patch = lds, *wl = lds + PATCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
    // m-tile m of a wave: row wave*2 + (m >> 1), pixels 16 * (m & 1) .. + 15
    const int a_base = ((wave * 2) * PW + (lane & 15)) * REC + (lane >> 4) * 4;
    const int b_base = ((lane >> 4) * 64 + (lane & 15)) * 4;
    STAMP_BEGIN();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) {
            const int ky = tt / 3, kx = tt % 3;
            f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float *rec = &patch[a_base + (((m >> 1) + ky) * PW + (m & 1) * 16 + kx) * REC];
                ah[m] = *(const f16x8 *)rec;
                al[m] = *(const f16x8 *)(rec + 16);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bh[n] = *(const f16x8 *)&wl[b_base + ((kx * 2 + 0) * 4 * 64 + n * 16) * 4];
                bl[n] = *(const f16x8 *)&wl[b_base + ((kx * 2 + 1) * 4 * 64 + n * 16) * 4];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
    }
    STAMP_END(st);
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------- V3: 16x16x32, filter in registers
__global__ __launch_bounds__(256, 1) void v3_kernel(const f16x8 *wreg, float *sink, Stamp *st, int iters, uint32_t seed) {
    constexpr int PW = 34, REC = 40, CH = 10 * PW * REC;  // one 32-channel patch buffer
    __shared__ __attribute__((aligned(16))) float lds[2 * CH];
    fill_lds((uint32_t *)lds, sizeof(lds) / 4, seed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = wave & 1, half = wave >> 1;
    f16x8 bh[2][9][2], bl[2][9][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[c][t][n] = wreg[((((c * 9 + t) * 2 + np) * 2 + n) * 2 + 0) * 64 + lane];
                bl[c][t][n] = wreg[((((c * 9 + t) * 2 + np) * 2 + n) * 2 + 1) * 64 + lane];
            }
    f32x4 acc[8][2];
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
    const int a_base = ((half * 4) * PW + (lane & 15)) * REC + (lane >> 4) * 4;
    STAMP_BEGIN();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) {
                const int ky = tt / 3, kx = tt % 3;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const float *rec = &lds[c * CH + a_base + (((m >> 1) + ky) * PW + (m & 1) * 16 + kx) * REC];
                    const f16x8 ah = *(const f16x8 *)rec, al = *(const f16x8 *)(rec + 16);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[c][tt][n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[c][tt][n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[c][tt][n], acc[m][n], 0, 0, 0);
                    }
                }
            }
    }
    STAMP_END(st);
    float s = 0.f;
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void fill_global(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = rnd_pair((uint32_t)i, seed);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7, per_round = 40;
    const int base = 256;  // V0 chunk-steps per workgroup; 512 workgroups x 4 waves x 108 MFMAs each
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *sink;
    Stamp *st;
    f16x8 *wreg;
    CK(hipMalloc(&sink, 2 * cus * 256 * sizeof(float)));
    CK(hipMalloc(&st, 2 * cus * sizeof(Stamp)));
    CK(hipMalloc(&wreg, 4 * 9 * 2 * 2 * 64 * sizeof(f16x8)));
    fill_global<<<64, 256>>>((uint32_t *)wreg, 4 * 9 * 2 * 2 * 64 * 4, 99u);
    CK(hipDeviceSynchronize());
    struct Var { const char *name; int wgs, iters; double mfma32; } v[4] = {
        {"V0 32x32x16  A+B from LDS   2 WG/CU (conv_mfma loop)", 2 * cus, base, 108.0},
        {"V1 32x32x16  B in registers 1 WG/CU", cus, base / 2, 432.0},
        {"V2 16x16x32  A+B from LDS   2 WG/CU (32-ch chunks)", 2 * cus, base / 2, 216.0},
        {"V3 16x16x32  B in registers 1 WG/CU", cus, base / 2, 432.0}};
    auto launch = [&](int k) {
        switch (k) {
            case 0: v0_kernel<<<v[0].wgs, 256>>>(sink, st, v[0].iters, 1u); break;
            case 1: v1_kernel<<<v[1].wgs, 256>>>(wreg, sink, st, v[1].iters, 1u); break;
            case 2: v2_kernel<<<v[2].wgs, 256>>>(sink, st, v[2].iters, 1u); break;
            case 3: v3_kernel<<<v[3].wgs, 256>>>(wreg, sink, st, v[3].iters, 1u); break;
        }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // warm-up: > 2 s of back-to-back launches so that the clock has settled under load
    for (int i = 0; i < 600; ++i) launch(i & 3);
    CK(hipDeviceSynchronize());
    std::vector<float> t[4];
    std::vector<double> mhz[4];
    std::vector<Stamp> hs(2 * cus);
    for (int r = 0; r < rounds; ++r)
        for (int k = 0; k < 4; ++k) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < per_round; ++i) launch(k);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t[k].push_back(ms / per_round);
            CK(hipMemcpy(hs.data(), st, v[k].wgs * sizeof(Stamp), hipMemcpyDeviceToHost));
            std::vector<double> c;
            for (int b = 0; b < v[k].wgs; ++b)
                if (hs[b].rt) c.push_back(100.0 * (double)hs[b].cyc / (double)hs[b].rt);
            std::sort(c.begin(), c.end());
            mhz[k].push_back(c.empty() ? 0.0 : c[c.size() / 2]);
        }
    printf("# %s, %d CUs; %d rounds x %d launches per variant, interleaved; equal matrix FLOPs per launch\n", prop.name, cus,
           rounds, per_round);
    printf("%-58s %9s %9s %9s %8s %12s\n", "variant", "median ms", "min ms", "TFLOP/s", "of 2500", "in-kernel MHz");
    for (int k = 0; k < 4; ++k) {
        std::sort(t[k].begin(), t[k].end());
        std::sort(mhz[k].begin(), mhz[k].end());
        const double flop = (double)v[k].wgs * 4 * v[k].iters * v[k].mfma32 * 32768.0;
        const double med = t[k][t[k].size() / 2];
        printf("%-58s %9.4f %9.4f %9.1f %8.3f %12.0f\n", v[k].name, med, t[k][0], flop / med / 1e9, flop / med / 1e9 / 2500.0,
               mhz[k][mhz[k].size() / 2]);
    }
    return 0;
}
