// resample.hip -- HBM-bound gather / resampling / layout kernels (NHWC, fp32).
//
//   warp        flow_warp / torch_warp            /root/reference/DCVC_HEM/src/models/video_net.py:32-55
//   up2         bilinearupsacling (*2.0 in SpyNet) video_net.py:58-63,139
//   down2       bilineardownsacling, avg_pool2d    video_net.py:66-71,132-133
//   maxpool2    nn.MaxPool2d(2)                    video_net.py:185
//   layout      NCHW <-> strided NHWC at the operator boundary
//
// All of these move each byte once; they are written for coalescing: a wave covers
// consecutive (pixel, channel) pairs of the NHWC tensor, 16 B per lane where the channel
// count allows it, so every wave-instruction touches whole 128-B lines.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dcvc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

#define RET_LAUNCH() return hipGetLastError() == hipSuccess ? DCVC_OK : DCVC_E_LAUNCH

inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

// torch.linspace(-1, 1, n)[i] in fp32 (symmetric form used by ATen's CPU kernel)
__device__ __forceinline__ float lin11(int i, int n) {
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

struct Tap {
    int x0, x1, y0, y1;
    float nw, ne, sw, se;
};

// Source coordinates exactly as the reference builds them: normalised grid + flow/((size-1)/2),
// un-normalised as (g + 1) * ((size-1)/2), clipped to the border (grid_sample, align_corners).
__device__ __forceinline__ Tap make_tap(float fx, float fy, int x, int y, int W, int H) {
    const float hx = (float)(((double)W - 1.0) / 2.0), hy = (float)(((double)H - 1.0) / 2.0);
    float gx = lin11(x, W) + fx / hx;
    float gy = lin11(y, H) + fy / hy;
    float ix = (gx + 1.0f) * hx, iy = (gy + 1.0f) * hy;
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
    Tap t;
    t.x0 = (int)xw;
    t.y0 = (int)yn;
    t.x1 = min(t.x0 + 1, W - 1);
    t.y1 = min(t.y0 + 1, H - 1);
    t.nw = s * e;
    t.ne = s * w;
    t.sw = n * e;
    t.se = n * w;
    return t;
}

__global__ void warp_vec4(const float *__restrict__ src, int src_cs, const float *__restrict__ flow, int flow_cs,
                          float *__restrict__ out, int out_cs, int N, int H, int W, int C4) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * H * W * C4;
    if (gid >= total) return;
    const int c4 = (int)(gid % C4);
    const int64_t pix = gid / C4;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    const float *fp = flow + pix * flow_cs;
    const Tap t = make_tap(fp[0], fp[1], x, y, W, H);
    const float *b = src + n * (int64_t)H * W * src_cs + c4 * 4;
    const f32x4 vnw = *(const f32x4 *)(b + ((int64_t)t.y0 * W + t.x0) * src_cs);
    const f32x4 vne = *(const f32x4 *)(b + ((int64_t)t.y0 * W + t.x1) * src_cs);
    const f32x4 vsw = *(const f32x4 *)(b + ((int64_t)t.y1 * W + t.x0) * src_cs);
    const f32x4 vse = *(const f32x4 *)(b + ((int64_t)t.y1 * W + t.x1) * src_cs);
    f32x4 r = vnw * t.nw + vne * t.ne + vsw * t.sw + vse * t.se;
    *(f32x4 *)(out + pix * out_cs + c4 * 4) = r;
}

// The wave-shuffle form for power-of-two channel groups: the G lanes that share a pixel (G = C/4 lanes of
// 4 channels each) need the same four tap weights and source offsets.  Only the group's first lane loads the
// flow vector and does the coordinate arithmetic (two divisions, two floors, the clamps); the other lanes
// receive the six results through ds_swizzle broadcasts inside their 16- or 32-lane row -- no LDS memory, no
// repeated flow loads -- and spend their cycles on the 4 x 16-byte gathers.
template <int G>
__global__ void warp_shfl(const float *__restrict__ src, int src_cs, const float *__restrict__ flow, int flow_cs,
                          float *__restrict__ out, int out_cs, int N, int H, int W) {
    static_assert(G >= 2 && G <= 32 && (G & (G - 1)) == 0, "lanes per pixel");
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * H * W * G;
    if (gid >= total) return;  // whole groups leave together (256 % G == 0)
    const int c4 = (int)(gid & (G - 1));
    const int64_t pix = gid / G;
    const int64_t n = pix / ((int64_t)W * H);
    float wnw = 0.f, wne = 0.f, wsw = 0.f, wse = 0.f;
    int off = 0, step = 0;
    if (c4 == 0) {
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const float *fp = flow + pix * flow_cs;
        const Tap t = make_tap(fp[0], fp[1], x, y, W, H);
        wnw = t.nw, wne = t.ne, wsw = t.sw, wse = t.se;
        off = t.y0 * W + t.x0;
        step = (t.x1 - t.x0) | ((t.y1 - t.y0) << 1);
    }
    // bit-mode swizzle: source lane = lane & ~(G - 1) within each half of the wave
    constexpr int PAT = 0x1F & ~(G - 1);
    wnw = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, wnw), PAT));
    wne = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, wne), PAT));
    wsw = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, wsw), PAT));
    wse = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, wse), PAT));
    off = __builtin_amdgcn_ds_swizzle(off, PAT);
    step = __builtin_amdgcn_ds_swizzle(step, PAT);
    const int64_t dx = (int64_t)(step & 1) * src_cs, dy = (int64_t)(step >> 1) * W * src_cs;
    const float *b = src + (n * (int64_t)H * W + off) * src_cs + c4 * 4;
    const f32x4 vnw = *(const f32x4 *)b;
    const f32x4 vne = *(const f32x4 *)(b + dx);
    const f32x4 vsw = *(const f32x4 *)(b + dy);
    const f32x4 vse = *(const f32x4 *)(b + dy + dx);
    // same expression as warp_vec4: bit-identical results
    const f32x4 r = vnw * wnw + vne * wne + vsw * wsw + vse * wse;
    *(f32x4 *)(out + pix * out_cs + c4 * 4) = r;
}

__global__ void warp_scalar(const float *__restrict__ src, int src_cs, const float *__restrict__ flow, int flow_cs,
                            float *__restrict__ out, int out_cs, int N, int H, int W, int C) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * H * W * C;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    const float *fp = flow + pix * flow_cs;
    const Tap t = make_tap(fp[0], fp[1], x, y, W, H);
    const float *b = src + n * (int64_t)H * W * src_cs + c;
    const float vnw = b[((int64_t)t.y0 * W + t.x0) * src_cs], vne = b[((int64_t)t.y0 * W + t.x1) * src_cs];
    const float vsw = b[((int64_t)t.y1 * W + t.x0) * src_cs], vse = b[((int64_t)t.y1 * W + t.x1) * src_cs];
    out[pix * out_cs + c] = vnw * t.nw + vne * t.ne + vsw * t.sw + vse * t.se;
}

// F.interpolate(bilinear, align_corners=False) to exactly twice the size, times scale.
__global__ void up2_kernel(const float *__restrict__ src, int src_cs, float *__restrict__ out, int out_cs,
                           float *__restrict__ out2, int out2_cs, int N, int H, int W, int C, float scale) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t total = (int64_t)N * Ho * Wo * C;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int64_t n = pix / ((int64_t)Wo * Ho);
    float sx = fmaxf(((float)ox + 0.5f) * 0.5f - 0.5f, 0.f), sy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.f);
    const int x0 = (int)sx, y0 = (int)sy;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float lx1 = sx - (float)x0, lx0 = 1.f - lx1, ly1 = sy - (float)y0, ly0 = 1.f - ly1;
    const float *b = src + n * (int64_t)H * W * src_cs + c;
    const float v00 = b[((int64_t)y0 * W + x0) * src_cs], v01 = b[((int64_t)y0 * W + x1) * src_cs];
    const float v10 = b[((int64_t)y1 * W + x0) * src_cs], v11 = b[((int64_t)y1 * W + x1) * src_cs];
    const float v = (ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11)) * scale;
    out[pix * out_cs + c] = v;
    if (out2) out2[pix * out2_cs + c] = v;
}

// mode 0: bilinear x0.5 (each output = l0y*(l0x*a + l1x*b) + l1y*(l0x*c + l1x*d), all 0.5);
// mode 1: avg_pool2d order (((a+b)+c)+d)/4; mode 2: max.  Result times scale (modes 0/1).
__global__ void down2_kernel(const float *__restrict__ src, int src_cs, float *__restrict__ out, int out_cs, int N,
                             int H, int W, int C, float scale, int mode) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo * C;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int64_t n = pix / ((int64_t)Wo * Ho);
    const float *b = src + ((n * H + 2 * oy) * (int64_t)W + 2 * ox) * src_cs + c;
    const float v00 = b[0], v01 = b[src_cs], v10 = b[(int64_t)W * src_cs], v11 = b[(int64_t)W * src_cs + src_cs];
    float v;
    if (mode == 0)
        v = (0.5f * (0.5f * v00 + 0.5f * v01) + 0.5f * (0.5f * v10 + 0.5f * v11)) * scale;
    else if (mode == 1)
        v = ((((v00 + v01) + v10) + v11) / 4.0f) * scale;
    else
        v = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
    out[pix * out_cs + c] = v;
}

__global__ void copy_channels_kernel(const float *__restrict__ src, int src_cs, float *__restrict__ out, int out_cs,
                                     int64_t npix, int C) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * C) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    out[pix * out_cs + c] = src[pix * src_cs + c];
}

// NCHW -> NHWC through an LDS tile so both sides are coalesced: a block handles 64 pixels of
// one image for all channels (C <= 64 per pass).
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ src, float *__restrict__ out, int out_cs, int C,
                                    int64_t HW) {
    __shared__ float tile[64][65];
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int64_t n = blockIdx.y;
    const int t = threadIdx.x;  // 256 threads
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int cc = min(64, C - c0);
        for (int i = t; i < cc * 64; i += 256) {
            const int c = i >> 6, p = i & 63;
            if (p0 + p < HW) tile[c][p] = src[(n * C + c0 + c) * HW + p0 + p];
        }
        __syncthreads();
        for (int i = t; i < cc * 64; i += 256) {
            const int p = i / cc, c = i - p * cc;
            if (p0 + p < HW) out[(n * HW + p0 + p) * out_cs + c0 + c] = tile[c][p];
        }
        __syncthreads();
    }
}

__global__ void nhwc_to_nchw_kernel(const float *__restrict__ src, int src_cs, float *__restrict__ out, int C,
                                    int64_t HW, int clamp01) {
    __shared__ float tile[64][65];
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int64_t n = blockIdx.y;
    const int t = threadIdx.x;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int cc = min(64, C - c0);
        for (int i = t; i < cc * 64; i += 256) {
            const int p = i / cc, c = i - p * cc;
            if (p0 + p < HW) tile[c][p] = src[(n * HW + p0 + p) * src_cs + c0 + c];
        }
        __syncthreads();
        for (int i = t; i < cc * 64; i += 256) {
            const int c = i >> 6, p = i & 63;
            if (p0 + p < HW) {
                float v = tile[c][p];
                if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
                out[(n * C + c0 + c) * HW + p0 + p] = v;
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int dcvc_warp(const float *src, int32_t src_cs, const float *flow, int32_t flow_cs, float *out,
                         int32_t out_cs, int32_t N, int32_t H, int32_t W, int32_t C, void *stream) {
    if (!src || !flow || !out || N <= 0 || H <= 1 || W <= 1 || C <= 0) return DCVC_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (C % 4 == 0) && (src_cs % 4 == 0) && (out_cs % 4 == 0) && !((uintptr_t)src & 15) &&
                     !((uintptr_t)out & 15);
    if (vec) {
        const int64_t total = (int64_t)N * H * W * (C / 4);
        const dim3 grid(nblk(total, 256)), blk(256);
        switch (C / 4) {  // wave-shuffle kernel for power-of-two channel groups, else every lane computes its tap
            case 2: hipLaunchKernelGGL(warp_shfl<2>, grid, blk, 0, st, src, src_cs, flow, flow_cs, out, out_cs, N, H, W); break;
            case 4: hipLaunchKernelGGL(warp_shfl<4>, grid, blk, 0, st, src, src_cs, flow, flow_cs, out, out_cs, N, H, W); break;
            case 8: hipLaunchKernelGGL(warp_shfl<8>, grid, blk, 0, st, src, src_cs, flow, flow_cs, out, out_cs, N, H, W); break;
            case 16: hipLaunchKernelGGL(warp_shfl<16>, grid, blk, 0, st, src, src_cs, flow, flow_cs, out, out_cs, N, H, W); break;
            case 32: hipLaunchKernelGGL(warp_shfl<32>, grid, blk, 0, st, src, src_cs, flow, flow_cs, out, out_cs, N, H, W); break;
            default:
                hipLaunchKernelGGL(warp_vec4, grid, blk, 0, st, src, src_cs, flow, flow_cs, out, out_cs, N, H, W, C / 4);
        }
    } else {
        const int64_t total = (int64_t)N * H * W * C;
        hipLaunchKernelGGL(warp_scalar, dim3(nblk(total, 256)), dim3(256), 0, st, src, src_cs, flow, flow_cs, out,
                           out_cs, N, H, W, C);
    }
    RET_LAUNCH();
}

extern "C" int dcvc_up2(const float *src, int32_t src_cs, float *out, int32_t out_cs, float *out2, int32_t out2_cs,
                        int32_t N, int32_t H, int32_t W, int32_t C, float scale, void *stream) {
    if (!src || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * 4 * H * W * C;
    hipLaunchKernelGGL(up2_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, src, src_cs, out, out_cs,
                       out2, out2_cs, N, H, W, C, scale);
    RET_LAUNCH();
}

static int down2_any(const float *src, int32_t src_cs, float *out, int32_t out_cs, int32_t N, int32_t H, int32_t W,
                     int32_t C, float scale, int mode, void *stream) {
    if (!src || !out || N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0) return DCVC_E_ARG;
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(down2_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, src, src_cs, out,
                       out_cs, N, H, W, C, scale, mode);
    RET_LAUNCH();
}

extern "C" int dcvc_down2(const float *src, int32_t src_cs, float *out, int32_t out_cs, int32_t N, int32_t H,
                          int32_t W, int32_t C, float scale, int32_t avgpool_order, void *stream) {
    return down2_any(src, src_cs, out, out_cs, N, H, W, C, scale, avgpool_order ? 1 : 0, stream);
}

extern "C" int dcvc_maxpool2(const float *src, int32_t src_cs, float *out, int32_t out_cs, int32_t N, int32_t H,
                             int32_t W, int32_t C, void *stream) {
    return down2_any(src, src_cs, out, out_cs, N, H, W, C, 1.f, 2, stream);
}

extern "C" int dcvc_copy_channels(const float *src, int32_t src_cs, float *out, int32_t out_cs, int64_t npix, int32_t C,
                                  void *stream) {
    if (!src || !out || npix <= 0 || C <= 0) return DCVC_E_ARG;
    hipLaunchKernelGGL(copy_channels_kernel, dim3(nblk(npix * C, 256)), dim3(256), 0, (hipStream_t)stream, src, src_cs,
                       out, out_cs, npix, C);
    RET_LAUNCH();
}

extern "C" int dcvc_nchw_to_nhwc(const float *src, float *out, int32_t out_cs, int32_t N, int32_t C, int32_t H,
                                 int32_t W, void *stream) {
    if (!src || !out || N <= 0 || C <= 0 || out_cs < C) return DCVC_E_ARG;
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblk(HW, 64), N), dim3(256), 0, (hipStream_t)stream, src, out, out_cs,
                       C, HW);
    RET_LAUNCH();
}

extern "C" int dcvc_nhwc_to_nchw(const float *src, int32_t src_cs, float *out, int32_t N, int32_t C, int32_t H,
                                 int32_t W, int32_t clamp01, void *stream) {
    if (!src || !out || N <= 0 || C <= 0 || src_cs < C) return DCVC_E_ARG;
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblk(HW, 64), N), dim3(256), 0, (hipStream_t)stream, src, src_cs, out,
                       C, HW, clamp01);
    RET_LAUNCH();
}
