// Bilinear resampling of an image batch under one affine map per sample, and its adjoint -- the geometric execution step of the ADA
// augmentation pipeline.
//
// Reference: `torch.nn.functional.affine_grid(theta, size, align_corners=False)` followed by `grid_sample_gradfix.grid_sample(images, grid)`
// (bilinear, zeros padding) in AugmentPipe.forward, src/training/augment.py:297-300, and the backward of that pair
// (grid_sample_gradfix.py:45-83).  The reference materialises the sampling grid ([N, Ho, Wo, 2] floats: 70 MB for 32 videos at 524^2) and
// reads it back in the sampler and again in its backward; here the grid is three FMAs per output pixel, evaluated in registers:
//
//   gather   y[n,c,Y,X]  = sum_{4 taps} w_tap(Y,X) * x[n,c,tap]                       one thread per output pixel, all channels
//   adjoint  dx[n,c,p]   = sum_{(Y,X): p is a tap of (Y,X)} w_tap(Y,X) * dy[n,c,Y,X]     one thread per SOURCE pixel p, again a gather: the output
//            pixels whose bilinear footprint contains p lie in the parallelogram M^-1 (p - t + (-1,1)^2) of the sample's affine map; the thread walks that
//            parallelogram's bounding box (3x3 .. 5x5 output pixels for the scales ADA draws), recomputes each candidate's taps with the forward's own
//            arithmetic and sums in a fixed order -- no atomics, deterministic.  (Round 3's form scattered with 36 atomics per output pixel, as ATen's
//            backward does: 4.3 ms per call at 32 videos, 6 ms of the aug=ada step; profiles/r04_c8_ada_step_kernel_stats.csv.)  A sample whose map is too
//            anisotropic / singular for a small box (bounding half-width > 6 output pixels) is left to the scatter kernel, which skips all others.
//
// with (ix, iy) = unnormalise(theta[n] @ (xn, yn, 1)),  xn = (2X + 1) / Wo - 1,  ix = ((gx + 1) * W - 1) / 2  -- the arithmetic of ATen's
// affine_grid / grid_sampler_2d for align_corners = False.  Both maps are linear in the image, so the pair serves every order of
// derivative w.r.t. the image (ops/resample.py).  HBM-bound: algorithmic bytes = 4 * N * C * (H * W + Ho * Wo).

#include "sgv_common.h"

#include <algorithm>
#include <stdlib.h>

namespace {

struct resample_params {
    const float* src;      // gather: x [n,c,h,w]; scatter: dy [n,c,ho,wo]
    float* dst;            // gather: y [n,c,ho,wo]; scatter: dx [n,c,h,w] (zero-initialised by the caller)
    const float* theta;    // [n, 2, 3]
    int n, c, h, w, ho, wo;
};

template <bool ADJOINT>
__global__ __launch_bounds__(256) void affine_resample_kernel(resample_params p) {
    const int X = blockIdx.x * 64 + (threadIdx.x & 63);
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (X >= p.wo || Y >= p.ho) return;
    const float* th = p.theta + (size_t)n * 6;
    const float xn = (2 * X + 1) / (float)p.wo - 1.f, yn = (2 * Y + 1) / (float)p.ho - 1.f;
    const float gx = th[0] * xn + th[1] * yn + th[2], gy = th[3] * xn + th[4] * yn + th[5];
    const float ix = ((gx + 1.f) * p.w - 1.f) * 0.5f, iy = ((gy + 1.f) * p.h - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    const bool vx0 = x0 >= 0 && x0 < p.w, vx1 = x0 + 1 >= 0 && x0 + 1 < p.w, vy0 = y0 >= 0 && y0 < p.h, vy1 = y0 + 1 >= 0 && y0 + 1 < p.h;
    const size_t plane_i = (size_t)p.h * p.w, plane_o = (size_t)p.ho * p.wo;
    const size_t o = (size_t)n * p.c * plane_o + (size_t)Y * p.wo + X;
    const ptrdiff_t i00 = (ptrdiff_t)y0 * p.w + x0;
    if (!ADJOINT) {
        const float* xb = p.src + (size_t)n * p.c * plane_i;
        for (int ch = 0; ch < p.c; ch++) {
            const float* q = xb + (size_t)ch * plane_i + i00;
            float v = 0.f;
            if (vy0 && vx0) v = __builtin_fmaf(w00, q[0], v);
            if (vy0 && vx1) v = __builtin_fmaf(w01, q[1], v);
            if (vy1 && vx0) v = __builtin_fmaf(w10, q[p.w], v);
            if (vy1 && vx1) v = __builtin_fmaf(w11, q[p.w + 1], v);
            p.dst[o + (size_t)ch * plane_o] = v;
        }
    } else {
        float* xb = p.dst + (size_t)n * p.c * plane_i;
        for (int ch = 0; ch < p.c; ch++) {
            const float g = p.src[o + (size_t)ch * plane_o];
            float* q = xb + (size_t)ch * plane_i + i00;
            if (vy0 && vx0) atomicAdd(q, w00 * g);
            if (vy0 && vx1) atomicAdd(q + 1, w01 * g);
            if (vy1 && vx0) atomicAdd(q + p.w, w10 * g);
            if (vy1 && vx1) atomicAdd(q + p.w + 1, w11 * g);
        }
    }
}

// half-widths of the bounding box of M^-1 [-1,1]^2 in output pixels (M: output pixel -> source pixel); false: singular / too large for the gather form
__device__ __forceinline__ bool adjoint_box(const resample_params& p, const float* th, float& a, float& b, float& d, float& e, float& hx, float& hy) {
    a = th[0] * p.w / (float)p.wo; b = th[1] * p.w / (float)p.ho;      // d(ix)/dX, d(ix)/dY
    d = th[3] * p.h / (float)p.wo; e = th[4] * p.h / (float)p.ho;      // d(iy)/dX, d(iy)/dY
    const float det = a * e - b * d;
    if (!(fabsf(det) > 1e-12f)) return false;
    hx = (fabsf(e) + fabsf(b)) / fabsf(det);
    hy = (fabsf(d) + fabsf(a)) / fabsf(det);
    return hx <= 6.f && hy <= 6.f;
}

constexpr int ADJ_MAXC = 12;    // channels per pass of the gather-form adjoint (ADA: 9 = 3 frames x RGB, or 3)

__global__ __launch_bounds__(256) void affine_resample_adjoint_gather_kernel(resample_params p) {
    const int xp = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yp = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    const float* th = p.theta + (size_t)n * 6;
    float a, b, d, e, hx, hy;
    if (!adjoint_box(p, th, a, b, d, e, hx, hy)) return;     // (sample-uniform: the scatter kernel takes this sample)
    if (xp >= p.w || yp >= p.h) return;
    // centre of the candidate box: the output position that maps onto p.  ix = a X + b Y + cx with cx from the forward's formula at X = Y = 0.
    const float cx = ((th[0] * (1.f / p.wo - 1.f) + th[1] * (1.f / p.ho - 1.f) + th[2] + 1.f) * p.w - 1.f) * 0.5f;
    const float cy = ((th[3] * (1.f / p.wo - 1.f) + th[4] * (1.f / p.ho - 1.f) + th[5] + 1.f) * p.h - 1.f) * 0.5f;
    const float det = a * e - b * d;
    const float qx = (e * (xp - cx) - b * (yp - cy)) / det, qy = (-d * (xp - cx) + a * (yp - cy)) / det;
    // one output pixel of slack on every side: the box is computed in different arithmetic than the taps
    const int X0 = max(0, (int)ceilf(qx - hx) - 1), X1 = min(p.wo - 1, (int)floorf(qx + hx) + 1);
    const int Y0 = max(0, (int)ceilf(qy - hy) - 1), Y1 = min(p.ho - 1, (int)floorf(qy + hy) + 1);
    const size_t plane_i = (size_t)p.h * p.w, plane_o = (size_t)p.ho * p.wo;
    const float* gb = p.src + (size_t)n * p.c * plane_o;
    float* ob = p.dst + (size_t)n * p.c * plane_i + (size_t)yp * p.w + xp;
    for (int c0 = 0; c0 < p.c; c0 += ADJ_MAXC) {
        const int nc = min(ADJ_MAXC, p.c - c0);
        float acc[ADJ_MAXC];
#pragma unroll
        for (int k = 0; k < ADJ_MAXC; k++) acc[k] = 0.f;
        for (int Y = Y0; Y <= Y1; Y++) {
            const float yn = (2 * Y + 1) / (float)p.ho - 1.f;
            for (int X = X0; X <= X1; X++) {
                // the forward kernel's own arithmetic for this output pixel: identical taps, an exact adjoint
                const float xn = (2 * X + 1) / (float)p.wo - 1.f;
                const float gx = th[0] * xn + th[1] * yn + th[2], gy = th[3] * xn + th[4] * yn + th[5];
                const float ix = ((gx + 1.f) * p.w - 1.f) * 0.5f, iy = ((gy + 1.f) * p.h - 1.f) * 0.5f;
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy;
                const float tx = ix - fx, ty = iy - fy;
                const float wx = xp == x0 ? 1.f - tx : (xp == x0 + 1 ? tx : 0.f);
                const float wy = yp == y0 ? 1.f - ty : (yp == y0 + 1 ? ty : 0.f);
                if (wx == 0.f || wy == 0.f || !(xp == x0 || xp == x0 + 1) || !(yp == y0 || yp == y0 + 1)) continue;
                const float wgt = wx * wy;      // (the forward multiplies the same two factors: (1 - tx) * (1 - ty), tx * (1 - ty), ...)
                const float* g = gb + (size_t)c0 * plane_o + (size_t)Y * p.wo + X;
#pragma unroll
                for (int k = 0; k < ADJ_MAXC; k++)
                    if (k < nc) acc[k] = __builtin_fmaf(wgt, g[(size_t)k * plane_o], acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < ADJ_MAXC; k++)
            if (k < nc) ob[(size_t)(c0 + k) * plane_i] = acc[k];
    }
}

// the atomics form of the adjoint for the samples the gather form leaves out
__global__ __launch_bounds__(256) void affine_resample_adjoint_scatter_rest_kernel(resample_params p) {
    const float* th = p.theta + (size_t)blockIdx.z * 6;
    float a, b, d, e, hx, hy;
    if (adjoint_box(p, th, a, b, d, e, hx, hy)) return;
    const int X = blockIdx.x * 64 + (threadIdx.x & 63);
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (X >= p.wo || Y >= p.ho) return;
    const float xn = (2 * X + 1) / (float)p.wo - 1.f, yn = (2 * Y + 1) / (float)p.ho - 1.f;
    const float gx = th[0] * xn + th[1] * yn + th[2], gy = th[3] * xn + th[4] * yn + th[5];
    const float ix = ((gx + 1.f) * p.w - 1.f) * 0.5f, iy = ((gy + 1.f) * p.h - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    const bool vx0 = x0 >= 0 && x0 < p.w, vx1 = x0 + 1 >= 0 && x0 + 1 < p.w, vy0 = y0 >= 0 && y0 < p.h, vy1 = y0 + 1 >= 0 && y0 + 1 < p.h;
    const size_t plane_i = (size_t)p.h * p.w, plane_o = (size_t)p.ho * p.wo;
    const size_t o = (size_t)n * p.c * plane_o + (size_t)Y * p.wo + X;
    float* xb = p.dst + (size_t)n * p.c * plane_i + (ptrdiff_t)y0 * p.w + x0;
    for (int ch = 0; ch < p.c; ch++) {
        const float g = p.src[o + (size_t)ch * plane_o];
        float* q = xb + (size_t)ch * plane_i;
        if (vy0 && vx0) atomicAdd(q, w00 * g);
        if (vy0 && vx1) atomicAdd(q + 1, w01 * g);
        if (vy1 && vx0) atomicAdd(q + p.w, w10 * g);
        if (vy1 && vx1) atomicAdd(q + p.w + 1, w11 * g);
    }
}

}  // namespace

extern "C" int sgv_affine_resample(const float* src, float* dst, const float* theta, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ho, int32_t wo,
                                   int32_t adjoint, void* stream_) {
    if (!src || !dst || !theta) return sgv_fail(SGV_ERR_INVALID_ARG, "affine_resample: NULL pointer");
    if (n < 1 || c < 1 || h < 1 || w < 1 || ho < 1 || wo < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "affine_resample: sizes must be positive");
    if (n > 65535 || (ho + 3) / 4 > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "affine_resample: batch / height too large");
    if ((int64_t)n * c * std::max((int64_t)h * w, (int64_t)ho * wo) > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "affine_resample: tensors are too large");
    hipStream_t stream = (hipStream_t)stream_;
    resample_params p{src, dst, theta, n, c, h, w, ho, wo};
    sgv_launch_scope scope(SGV_K_POINTWISE, stream, 4.0 * n * c * ((double)h * w + (double)ho * wo));
    dim3 grid((unsigned)((wo + 63) / 64), (unsigned)((ho + 3) / 4), (unsigned)n);
    if (!adjoint) {
        hipLaunchKernelGGL(affine_resample_kernel<false>, grid, dim3(256), 0, stream, p);
        return sgv_check_launch("affine_resample_kernel");
    }
    // adjoint (dst zero-initialised by the caller): the gather form over the source pixels; then the atomics form, which returns at once for every sample
    // the gather form served.  SGV_RESAMPLE_ADJOINT=scatter: round 3's all-atomics kernel.
    static const bool scatter_only = getenv("SGV_RESAMPLE_ADJOINT") && getenv("SGV_RESAMPLE_ADJOINT")[0] == 's';
    if (scatter_only || (h + 3) / 4 > 65535) {
        hipLaunchKernelGGL(affine_resample_kernel<true>, grid, dim3(256), 0, stream, p);
        return sgv_check_launch("affine_resample_kernel (adjoint, atomics)");
    }
    dim3 sgrid((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)n);
    hipLaunchKernelGGL(affine_resample_adjoint_gather_kernel, sgrid, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(affine_resample_adjoint_scatter_rest_kernel, grid, dim3(256), 0, stream, p);
    return sgv_check_launch("affine_resample_adjoint_gather_kernel");
}
