// Probe of ds_read_b64_tr_b16 (gfx950): which element lands where.  A [64 px][32 ch] fp16 image with
// value = px * 100 + ch; every lane reads with the operand addressing the weight-gradient kernel would
// use for A[row = ch][k = px] of v_mfma_f32_32x32x16_f16 and stores its 8 values.
#include <hip/hip_runtime.h>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float *out) {
    __shared__ __attribute__((aligned(16))) __fp16 img[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) img[i] = (__fp16)(float)((i / 32) * 100 + (i % 32));
    __syncthreads();
    const int l = threadIdx.x, j = l & 15, q = j >> 2, p = j & 3, h = l >> 5, g = (l >> 4) & 1;
    for (int rd = 0; rd < 2; ++rd) {
        const __fp16 *addr = img + (8 * h + 4 * rd + q) * 32 + 16 * g + 4 * p;
        h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4 *)addr);
        for (int e = 0; e < 4; ++e) out[l * 8 + rd * 4 + e] = (float)v[e];
    }
}
extern "C" int probe(float *out, void *stream) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
