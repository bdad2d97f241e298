// Probe: the flow-gradient part of warp_bwd_kernel as it was before round 2's change (reduction of the per-channel
// partials through LDS: write, barrier, lane 0 sums), next to the shuffle version, to look for the run-to-run
// difference seen beside the weight-gradient stream.  Not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ float lin11(int i, int n) {
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

template <int VARIANT>
__global__ void warp_dflow_kernel(const float *__restrict__ src, int src_cs, const float *__restrict__ flow, int flow_cs,
                                  const float *__restrict__ dout, int dout_cs, float *__restrict__ dflow, int dflow_cs,
                                  int N, int H, int W, int C, unsigned long long *__restrict__ fix) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.y + threadIdx.y;
    const bool live = pix < (int64_t)N * H * W;
    float gx_acc = 0.f, gy_acc = 0.f;
    if (live) {
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const int64_t n = pix / ((int64_t)W * H);
        const float fx = flow[pix * flow_cs], fy = flow[pix * flow_cs + 1];
        const float hx = (float)(((double)W - 1.0) / 2.0), hy = (float)(((double)H - 1.0) / 2.0);
        float ix = (lin11(x, W) + fx / hx + 1.0f) * hx, iy = (lin11(y, H) + fy / hy + 1.0f) * hy;
        const float mx = (ix <= 0.f || ix >= (float)(W - 1)) ? 0.f : 1.f;
        const float my = (iy <= 0.f || iy >= (float)(H - 1)) ? 0.f : 1.f;
        ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
        iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
        const float xw = floorf(ix), yn = floorf(iy);
        const float w = ix - xw, e = 1.0f - w, nn = iy - yn, s = 1.0f - nn;
        const int x0 = (int)xw, y0 = (int)yn, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const bool x1in = x0 + 1 <= W - 1, y1in = y0 + 1 <= H - 1;
        const size_t b = (size_t)n * H * W;
        const size_t pnw = b + (size_t)y0 * W + x0, pne = b + (size_t)y0 * W + x1;
        const size_t psw = b + (size_t)y1 * W + x0, pse = b + (size_t)y1 * W + x1;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float g = dout[pix * dout_cs + c];
            const float vnw = src[pnw * src_cs + c];
            const float vne = x1in ? src[pne * src_cs + c] : 0.f;
            const float vsw = y1in ? src[psw * src_cs + c] : 0.f;
            const float vse = (x1in && y1in) ? src[pse * src_cs + c] : 0.f;
            gx_acc += g * (s * (vne - vnw) + nn * (vse - vsw));
            gy_acc += g * (e * (vsw - vnw) + w * (vse - vne));
            if (fix) {
                atomicAdd(&fix[pnw * C + c], (unsigned long long)__float2ll_rn(g * (s * e) * 68719476736.f));
                if (x1in) atomicAdd(&fix[pne * C + c], (unsigned long long)__float2ll_rn(g * (s * w) * 68719476736.f));
                if (y1in) atomicAdd(&fix[psw * C + c], (unsigned long long)__float2ll_rn(g * (nn * e) * 68719476736.f));
                if (x1in && y1in) atomicAdd(&fix[pse * C + c], (unsigned long long)__float2ll_rn(g * (nn * w) * 68719476736.f));
            }
        }
        gx_acc *= mx;
        gy_acc *= my;
    }
    if (VARIANT == 4 || VARIANT == 5) {  // 4: volatile LDS accesses (no write2 / read2 merging); 5: gx, gy copied through v_mov first
        __shared__ float sm[512];
        volatile float *sx = sm, *sy = sm + 256;
        const int t = threadIdx.y * blockDim.x + threadIdx.x;
        if (VARIANT == 5) {
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\ts_nop 4" : "+v"(gx_acc), "+v"(gy_acc));
        }
        sx[t] = gx_acc;
        sy[t] = gy_acc;
        __syncthreads();
        if (threadIdx.x == 0 && live) {
            float ax = 0.f, ay = 0.f;
            for (int k = 0; k < (int)blockDim.x; ++k) {
                ax += sx[t + k];
                ay += sy[t + k];
            }
            dflow[pix * dflow_cs] += ax;
            dflow[pix * dflow_cs + 1] += ay;
        }
    } else if (VARIANT == 0 || VARIANT == 2 || VARIANT == 3) {  // 0: as shipped in round 1; 2: 4 KiB of unused LDS in front; 3: sy first
        __shared__ float sm[VARIANT == 2 ? 1536 : 512];
        float *sx = VARIANT == 2 ? sm + 1024 : (VARIANT == 3 ? sm + 256 : sm);
        float *sy = VARIANT == 2 ? sm + 1280 : (VARIANT == 3 ? sm : sm + 256);
        if (VARIANT == 2 && threadIdx.x == 0 && threadIdx.y == 0 && dflow == nullptr) sm[0] = 1.f;  // keep the pad allocated
        const int t = threadIdx.y * blockDim.x + threadIdx.x;
        sx[t] = gx_acc;
        sy[t] = gy_acc;
        __syncthreads();
        if (threadIdx.x == 0 && live) {
            float ax = 0.f, ay = 0.f;
            for (int k = 0; k < (int)blockDim.x; ++k) {
                ax += sx[t + k];
                ay += sy[t + k];
            }
            dflow[pix * dflow_cs] += ax;
            dflow[pix * dflow_cs + 1] += ay;
        }
    } else {             // round 2
        for (int m = (int)blockDim.x >> 1; m >= 1; m >>= 1) {
            gx_acc += __shfl_xor(gx_acc, m);
            gy_acc += __shfl_xor(gy_acc, m);
        }
        if (threadIdx.x == 0 && live) {
            dflow[pix * dflow_cs] += gx_acc;
            dflow[pix * dflow_cs + 1] += gy_acc;
        }
    }
}

extern "C" int probe_warp_dflow(int variant, const float *src, int src_cs, const float *flow, int flow_cs, const float *dout,
                                int dout_cs, float *dflow, int dflow_cs, int N, int H, int W, int C, void *fix, void *stream) {
    int Cl = 1;
    while (Cl * 2 <= C && Cl < 64) Cl *= 2;
    const int64_t npix = (int64_t)N * H * W;
    dim3 block(Cl, 256 / Cl), grid((unsigned)((npix + 256 / Cl - 1) / (256 / Cl)));
    if (variant == 4)
        hipLaunchKernelGGL(warp_dflow_kernel<4>, grid, block, 0, (hipStream_t)stream, src, src_cs, flow, flow_cs, dout, dout_cs,
                           dflow, dflow_cs, N, H, W, C, (unsigned long long *)fix);
    else if (variant == 5)
        hipLaunchKernelGGL(warp_dflow_kernel<5>, grid, block, 0, (hipStream_t)stream, src, src_cs, flow, flow_cs, dout, dout_cs,
                           dflow, dflow_cs, N, H, W, C, (unsigned long long *)fix);
    else if (variant == 2)
        hipLaunchKernelGGL(warp_dflow_kernel<2>, grid, block, 0, (hipStream_t)stream, src, src_cs, flow, flow_cs, dout, dout_cs,
                           dflow, dflow_cs, N, H, W, C, (unsigned long long *)fix);
    else if (variant == 3)
        hipLaunchKernelGGL(warp_dflow_kernel<3>, grid, block, 0, (hipStream_t)stream, src, src_cs, flow, flow_cs, dout, dout_cs,
                           dflow, dflow_cs, N, H, W, C, (unsigned long long *)fix);
    else if (variant == 0)
        hipLaunchKernelGGL(warp_dflow_kernel<0>, grid, block, 0, (hipStream_t)stream, src, src_cs, flow, flow_cs, dout, dout_cs,
                           dflow, dflow_cs, N, H, W, C, (unsigned long long *)fix);
    else
        hipLaunchKernelGGL(warp_dflow_kernel<1>, grid, block, 0, (hipStream_t)stream, src, src_cs, flow, flow_cs, dout, dout_cs,
                           dflow, dflow_cs, N, H, W, C, (unsigned long long *)fix);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
