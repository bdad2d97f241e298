// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950?  (one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float aval, float bval) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)aval; b[j] = (_Float16)bval; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float *d; hipMalloc(&d, 4);
    const float tests[][2] = {{9.5367431640625e-07f, 1.0f}, {5.9604644775390625e-08f, 1.0f}, {1.0f, 9.5367431640625e-07f},
                              {3.0517578125e-05f, 3.0517578125e-05f}, {0.5f, 0.25f}};
    for (auto &t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> %g (exact %g)\n", t[0], t[1], h, 16.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
    }
    return 0;
}
