// Developer probe (NOT part of libdcvc_hip.so): conv_k32 with s_memtime stamps at its phase boundaries, written by
// thread 0 of every workgroup to a buffer of 64 slots per workgroup (slot 62 / 63: s_memrealtime at entry / exit).
// Build: make -C tools/probes ; used by tools/conv_k32_stamps.py
#include <hip/hip_runtime.h>
__device__ unsigned long long *g_k32_stamps;
#define K32_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        if (threadIdx.x == 0) {                                                                                        \
            unsigned long long *s_ = g_k32_stamps + (size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64; \
            s_[(i)] = __builtin_amdgcn_s_memtime();                                                                    \
            if ((i) == 0) s_[62] = __builtin_amdgcn_s_memrealtime();                                                   \
            if ((i) == 59) s_[63] = __builtin_amdgcn_s_memrealtime();                                                  \
            if ((i) == 10) s_[61] = __builtin_amdgcn_s_memrealtime();                                                  \
        }                                                                                                              \
    } while (0)
#include "conv_k32_dev.hip"  // generated: product kernel + conv_k32_dev_switches.patch (Makefile)

extern "C" int k32_stamps_set(unsigned long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_k32_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
