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
#include <type_traits>
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


// ---------------------------------------------------------------------------------------------------------------------------------------------
// The whole geometric execution of the ADA pipeline as ONE kernel (forward direction):
//
//     y = down2( resample_theta( up2( reflect_pad(x, margin) ) ) )          src/training/augment.py:270-300
//
// with the 12-tap filter both FIR steps use (`Hz_geom`, sym6; up: gain 4, zero phase padding [6,5]; down: `flip_filter`, padding -2 * (12 / 4)).  The
// four-pass composition moves 1.6 GB per 32 clips (reflect pad 75 -> 83 MB, up-sampled 331 MB, resampled 316 MB, written and read back once each; 1.04 ms at the rates
// its kernels reach); the algorithm's own bytes are x in, y out: 150 MB.  A workgroup owns a 16 x 16 tile of y for one sample and walks the channels:
//   1. the 42 x 42 block of resampled ("hi-res") pixels the tile's down-sampling FIR reads maps, under the sample's affine map, onto a parallelogram of the
//      up-sampled padded image; its bounding box (<= GEO_BMAX^2: rotations by any angle at scales <= 1.15) fixes the box of PADDED-image pixels the up-sampling FIR
//      needs ((box / 2 + 6)^2), which is read from x with the reflection resolved in the index (the padded image never exists; positions outside it are zeros, as
//      upfirdn2d's own zero padding makes them);
//   2. up-sampling, separable, in LDS (one thread makes the even and the odd output of an input position: 7 LDS reads for 12 FMAs);
//   3. the bilinear taps with `affine_resample_kernel`'s own arithmetic (same taps, same weights), from LDS;
//   4. down-sampling, separable, in LDS; 256 stores.
// A sample whose box does not fit at 16 x 16 (zoom-out by more than ~1.15 at 45 degrees, 1.6 axis-aligned: a sixth of the maps ADA draws) is served in 8 x 8 or 4 x 4
// sub-tiles by the same code (`ada_geometric_forward_sub_kernel`, a second launch whose workgroups return at once for every other sample: 32 drawn maps 0.73 ms
// instead of 3.2 ms, profiles/r06_c37_*); beyond a zoom-out of ~4 the direct form of step 1-3: each hi-res pixel evaluates its four up-sampled taps from x
// (4 x 36 FMAs through L1 / L2) -- 20x the arithmetic, any map.
// The margin arrives by value: the batch's measured margin (host-side parameters) or the static worst case w - 1 / h - 1 (hipGraph capture) -- here it costs nothing
// either way, the padded image being virtual.  The backward pass: `ada_geometric_adjoint_kernel` below.

constexpr int GEO_TO = 16;                    // output tile
constexpr int GEO_HI = 2 * GEO_TO + 10;       // hi-res rows / columns a tile's down-sampling reads
constexpr int GEO_BMAX = 72;                  // largest staged box of up-sampled pixels (even)
constexpr int GEO_PMAX = GEO_BMAX / 2 + 6;    // ... and of padded-image pixels under it

struct geom_params {
    const float* x;
    float* y;
    const float* theta;                       // [n, 2, 3]: hi-res (normalised) -> up-sampled padded image (normalised), what affine_resample gets
    int n, c, h, w;
    int mx0, mx1, my0, my1;                   // reflect-padding margin
    float fe[6], fo[6];                       // up-sampling phases: even output 2i = sum_t fe[t] P[i + t - 3], odd output 2i + 1 = sum_t fo[t] P[i + t - 2]  (gain 2 folded)
    float fd[12];                             // down-sampling taps: out[o] = sum_m fd[m] hi[2o + 1 + m]
};

__device__ __forceinline__ int geo_reflect(int r, int n) { r = r < 0 ? -r : r; return r >= n ? 2 * (n - 1) - r : r; }

// value of the padded image at (qy, qx) in padded coordinates: reflected source pixel, or zero outside the padded extent
__device__ __forceinline__ float geo_padded(const geom_params& p, const float* plane, int qy, int qx) {
    if (qx < 0 || qx >= p.w + p.mx0 + p.mx1 || qy < 0 || qy >= p.h + p.my0 + p.my1) return 0.f;
    return plane[(size_t)geo_reflect(qy - p.my0, p.h) * p.w + geo_reflect(qx - p.mx0, p.w)];
}

// one up-sampled pixel straight from x (direct form)
__device__ float geo_upsampled_direct(const geom_params& p, const float* plane, int uy, int ux) {
    const int wu = 2 * (p.w + p.mx0 + p.mx1), hu = 2 * (p.h + p.my0 + p.my1);
    if (ux < 0 || ux >= wu || uy < 0 || uy >= hu) return 0.f;
    const bool ox = ux & 1, oy = uy & 1;
    const int bx = (ux >> 1) - 3 + (ux & 1), by = (uy >> 1) - 3 + (uy & 1);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        float row = 0.f;
#pragma unroll
        for (int t = 0; t < 6; t++) row = __builtin_fmaf(ox ? p.fo[t] : p.fe[t], geo_padded(p, plane, by + r, bx + t), row);
        acc = __builtin_fmaf(oy ? p.fo[r] : p.fe[r], row, acc);
    }
    return acc;
}

// one hi-res pixel in the direct form: the four bilinear taps, each an up-sampled pixel evaluated from x
__device__ float geo_hi_direct(const geom_params& p, const float* plane, float ix, float iy) {
    const float lim = 1e7f;
    ix = fminf(fmaxf(ix, -lim), lim); iy = fminf(fmaxf(iy, -lim), lim);
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    const int x0 = (int)fx, y0 = (int)fy;
    float v = 0.f;
    v = __builtin_fmaf((1.f - tx) * (1.f - ty), geo_upsampled_direct(p, plane, y0, x0), v);
    v = __builtin_fmaf(tx * (1.f - ty), geo_upsampled_direct(p, plane, y0, x0 + 1), v);
    v = __builtin_fmaf((1.f - tx) * ty, geo_upsampled_direct(p, plane, y0 + 1, x0), v);
    v = __builtin_fmaf(tx * ty, geo_upsampled_direct(p, plane, y0 + 1, x0 + 1), v);
    return v;
}

// one TO x TO tile of y (TO = 16: the kernel's tile; 8, 4: sub-tiles for samples whose map spreads a 16-tile's footprint beyond the staging buffers)
template <int TO>
__device__ __forceinline__ void geo_forward_tile(const geom_params& p, const int n, const int ox0, const int oy0, float* s_pin, float* s_t, float* s_u) {
    constexpr int GEO_TO = TO, GEO_HI = 2 * TO + 10;
    float* s_hi = s_t;                                      // [GEO_HI][GEO_HI + 1]
    float* s_dh = s_t + GEO_HI * (GEO_HI + 1);              // [GEO_HI][GEO_TO]
    static_assert(GEO_HI * (GEO_HI + 1) + GEO_HI * GEO_TO <= GEO_PMAX * GEO_BMAX, "the hi-res block shares the row buffer");

    const int tid = threadIdx.x;
    const int wp = p.w + p.mx0 + p.mx1, hp = p.h + p.my0 + p.my1, wu = 2 * wp, hu = 2 * hp;
    const int wo = 2 * (p.w + 6), ho = 2 * (p.h + 6);
    const int X0 = 2 * ox0 + 1, Y0 = 2 * oy0 + 1;           // first hi-res column / row of the tile
    const float* th = p.theta + (size_t)n * 6;
    const float t0 = th[0], t1 = th[1], t2 = th[2], t3 = th[3], t4 = th[4], t5 = th[5];

    // affine_resample_kernel's arithmetic for one hi-res pixel
    auto source = [&](int X, int Y, float& ix, float& iy) {
        const float xn = (2 * X + 1) / (float)wo - 1.f, yn = (2 * Y + 1) / (float)ho - 1.f;
        const float gx = t0 * xn + t1 * yn + t2, gy = t3 * xn + t4 * yn + t5;
        ix = ((gx + 1.f) * wu - 1.f) * 0.5f;
        iy = ((gy + 1.f) * hu - 1.f) * 0.5f;
    };
    // bounding box of the taps of the tile's hi-res block (extremes of an affine map sit on the corners; one pixel of slack for rounding)
    float cx[4], cy[4];
    source(X0, Y0, cx[0], cy[0]);
    source(X0 + GEO_HI - 1, Y0, cx[1], cy[1]);
    source(X0, Y0 + GEO_HI - 1, cx[2], cy[2]);
    source(X0 + GEO_HI - 1, Y0 + GEO_HI - 1, cx[3], cy[3]);
    const float lim = 1e7f;
    const float xmin = fmaxf(-lim, fminf(fminf(cx[0], cx[1]), fminf(cx[2], cx[3]))), xmax = fminf(lim, fmaxf(fmaxf(cx[0], cx[1]), fmaxf(cx[2], cx[3])));
    const float ymin = fmaxf(-lim, fminf(fminf(cy[0], cy[1]), fminf(cy[2], cy[3]))), ymax = fminf(lim, fmaxf(fmaxf(cy[0], cy[1]), fmaxf(cy[2], cy[3])));
    const int ux0 = ((int)floorf(xmin) - 1) & ~1, uy0 = ((int)floorf(ymin) - 1) & ~1;       // even: an input position makes the box's columns 2k, 2k + 1
    const int bw = (((int)floorf(xmax) + 2 - ux0 + 1) + 1) & ~1, bh = (((int)floorf(ymax) + 2 - uy0 + 1) + 1) & ~1;
    const bool staged = bw <= GEO_BMAX && bh <= GEO_BMAX && bw > 0 && bh > 0 && xmin > -lim && ymin > -lim && xmax < lim && ymax < lim;
    const int pw = bw / 2 + 6, ph = bh / 2 + 6;             // padded-image box under the up-sampled box
    const int px0 = (ux0 >> 1) - 3, py0 = (uy0 >> 1) - 3;

    // this thread's hi-res pixels (the same for every channel): LDS offset of the first tap and the two fractions; offset < 0: outside the hi-res image -> 0
    constexpr int PER = (GEO_HI * GEO_HI + 255) / 256;
    int s_off[PER];
    float s_tx[PER], s_ty[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int s = tid + k * 256;
        s_off[k] = -1; s_tx[k] = 0.f; s_ty[k] = 0.f;
        if (s < GEO_HI * GEO_HI) {
            const int ly = s / GEO_HI, lx = s - ly * GEO_HI;
            if (X0 + lx < wo && Y0 + ly < ho) {
                float ix, iy;
                source(X0 + lx, Y0 + ly, ix, iy);
                const float fx = floorf(ix), fy = floorf(iy);
                s_tx[k] = ix - fx; s_ty[k] = iy - fy;
                if (staged) s_off[k] = ((int)fy - uy0) * (GEO_BMAX + 1) + ((int)fx - ux0);
            }
        }
    }

    for (int ch = 0; ch < p.c; ch++) {
        const float* plane = p.x + ((size_t)n * p.c + ch) * p.h * p.w;
        if (staged) {
            // 1. padded-image box (reflection in the index, zeros outside the padded extent)
            for (int e = tid; e < ph * pw; e += 256) {
                const int r = e / pw, cidx = e - r * pw;
                s_pin[r * (GEO_PMAX + 1) + cidx] = geo_padded(p, plane, py0 + r, px0 + cidx);
            }
            __syncthreads();
            // 2a. up-sample along x: input position k of row r makes columns 2k and 2k + 1
            for (int e = tid; e < ph * (bw / 2); e += 256) {
                const int r = e / (bw / 2), k = e - r * (bw / 2);
                const float* q = s_pin + r * (GEO_PMAX + 1) + k;
                float v[7];
#pragma unroll
                for (int t = 0; t < 7; t++) v[t] = q[t];
                float ev = 0.f, od = 0.f;
#pragma unroll
                for (int t = 0; t < 6; t++) { ev = __builtin_fmaf(p.fe[t], v[t], ev); od = __builtin_fmaf(p.fo[t], v[t + 1], od); }
                s_t[r * GEO_BMAX + 2 * k] = ev;
                s_t[r * GEO_BMAX + 2 * k + 1] = od;
            }
            __syncthreads();
            // 2b. ... along y; zeros outside the up-sampled image (upfirdn2d crops there)
            for (int e = tid; e < (bh / 2) * bw; e += 256) {
                const int k = e / bw, j = e - k * bw;
                float v[7];
#pragma unroll
                for (int t = 0; t < 7; t++) v[t] = s_t[(k + t) * GEO_BMAX + j];
                float ev = 0.f, od = 0.f;
#pragma unroll
                for (int t = 0; t < 6; t++) { ev = __builtin_fmaf(p.fe[t], v[t], ev); od = __builtin_fmaf(p.fo[t], v[t + 1], od); }
                const bool inx = ux0 + j >= 0 && ux0 + j < wu;
                const int uy = uy0 + 2 * k;
                s_u[(2 * k) * (GEO_BMAX + 1) + j] = inx && uy >= 0 && uy < hu ? ev : 0.f;
                s_u[(2 * k + 1) * (GEO_BMAX + 1) + j] = inx && uy + 1 >= 0 && uy + 1 < hu ? od : 0.f;
            }
            __syncthreads();
        }
        // 3. bilinear taps -> hi-res block
        if (staged) {
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int s = tid + k * 256;
                if (s < GEO_HI * GEO_HI) {
                    const int ly = s / GEO_HI, lx = s - ly * GEO_HI;
                    float v = 0.f;
                    if (s_off[k] >= 0) {
                        const float* q = s_u + s_off[k];
                        const float tx = s_tx[k], ty = s_ty[k];
                        v = __builtin_fmaf((1.f - tx) * (1.f - ty), q[0], v);
                        v = __builtin_fmaf(tx * (1.f - ty), q[1], v);
                        v = __builtin_fmaf((1.f - tx) * ty, q[GEO_BMAX + 1], v);
                        v = __builtin_fmaf(tx * ty, q[GEO_BMAX + 2], v);
                    }
                    s_hi[ly * (GEO_HI + 1) + lx] = v;
                }
            }
        } else {
#pragma unroll 1
            for (int s = tid; s < GEO_HI * GEO_HI; s += 256) {
                const int ly = s / GEO_HI, lx = s - ly * GEO_HI;
                float v = 0.f;
                if (X0 + lx < wo && Y0 + ly < ho) {
                    float ix, iy;
                    source(X0 + lx, Y0 + ly, ix, iy);
                    v = geo_hi_direct(p, plane, ix, iy);
                }
                s_hi[ly * (GEO_HI + 1) + lx] = v;
            }
        }
        __syncthreads();
        // 4a. down-sample along x
        for (int e = tid; e < GEO_HI * GEO_TO; e += 256) {
            const int r = e / GEO_TO, o = e - r * GEO_TO;
            const float* q = s_hi + r * (GEO_HI + 1) + 2 * o;
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 12; m++) acc = __builtin_fmaf(p.fd[m], q[m], acc);
            s_dh[r * GEO_TO + o] = acc;
        }
        __syncthreads();
        // 4b. ... along y, store
        if (tid < GEO_TO * GEO_TO) {
            const int oy = tid / GEO_TO, ox = tid - oy * GEO_TO;
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 12; m++) acc = __builtin_fmaf(p.fd[m], s_dh[(2 * oy + m) * GEO_TO + ox], acc);
            if (ox0 + ox < p.w && oy0 + oy < p.h) p.y[((size_t)n * p.c + ch) * p.h * p.w + (size_t)(oy0 + oy) * p.w + ox0 + ox] = acc;
        }
        __syncthreads();      // s_dh / s_hi are rewritten by the next channel's step 2a
    }
}



// SUB = false: the samples served in 16 x 16 tiles (every map near the identity); SUB = true: the others.  Two launches, a sample's workgroups return at once
// from the one that does not serve it (one kernel holding both paths needs 145 registers: 10 % slower on the common path).
template <bool SUB>
__device__ __forceinline__ void geo_forward_body(const geom_params& p) {
    __shared__ float s_pin[GEO_PMAX * (GEO_PMAX + 1)];
    __shared__ float s_t[GEO_PMAX * GEO_BMAX];            // horizontally up-sampled rows; later the hi-res block + its horizontally down-sampled form
    __shared__ float s_u[GEO_BMAX * (GEO_BMAX + 1)];
    const int n = blockIdx.z;
    const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * 16;
    // the tile size whose footprint in the up-sampled image fits the staging buffers for this sample's map (the footprint of a (2 t + 10)^2 hi-res block is a
    // parallelogram; its bounding box grows with |a| + |b| up-sampled pixels per hi-res pixel): 16, else 8 x 8 or 4 x 4 sub-tiles (zoom-out up to ~2.7 / ~4 at any
    // angle); a sample beyond that keeps the 16-tile and its direct form
    const float* th = p.theta + (size_t)n * 6;
    const float wu = 2.f * (p.w + p.mx0 + p.mx1), hu = 2.f * (p.h + p.my0 + p.my1), wo = 2.f * (p.w + 6), ho = 2.f * (p.h + 6);
    const float sx = fabsf(th[0] * wu / wo) + fabsf(th[1] * wu / ho), sy = fabsf(th[3] * hu / wo) + fabsf(th[4] * hu / ho);
    const float s = fmaxf(sx, sy);
    int t = 16;
    if (!(41.f * s + 6.f <= (float)GEO_BMAX)) t = 25.f * s + 6.f <= (float)GEO_BMAX ? 8 : (17.f * s + 6.f <= (float)GEO_BMAX ? 4 : 16);
    if (!SUB) {
        if (t == 16) geo_forward_tile<16>(p, n, ox0, oy0, s_pin, s_t, s_u);
        return;
    }
    if (t == 16) return;
    for (int sub = 0; sub < (16 / t) * (16 / t); sub++) {
        const int sx0 = ox0 + (sub % (16 / t)) * t, sy0 = oy0 + (sub / (16 / t)) * t;
        if (sx0 >= p.w || sy0 >= p.h) continue;           // (uniform)
        if (t == 8) geo_forward_tile<8>(p, n, sx0, sy0, s_pin, s_t, s_u);
        else geo_forward_tile<4>(p, n, sx0, sy0, s_pin, s_t, s_u);
    }
}

__global__ __launch_bounds__(256, 4) void ada_geometric_forward_kernel(geom_params p) { geo_forward_body<false>(p); }
__global__ __launch_bounds__(256) void ada_geometric_forward_sub_kernel(geom_params p) { geo_forward_body<true>(p); }

// ---------------------------------------------------------------------------------------------------------------------------------------------
// ... and its ADJOINT as one kernel:   dx = pad^T( up2^T( resample^T( down2^T( dy ) ) ) )
//
// the gradient of the block w.r.t. the image (the generator's phase differentiates through augmented fakes, the R1 penalty twice through augmented reals:
// src/training/loss.py:91-110, :144-164; reference backward: autograd through F.pad, upfirdn2d.py:249-260, grid_sample_gradfix.py:45-83).  The block is linear in
// the image, so this kernel and the forward kernel are each other's derivative to every order (ops/resample.py `_AdaGeometric`).
//
// A workgroup (256 threads, three per CU) owns a 16 x 16 tile of dx for one sample.  A source pixel p appears in the (virtual) padded image up to 3 x 3 times -- itself, and its
// reflections about the left / right / top / bottom edges where the margin reaches that far -- so the tile is the sum over <= 9 IMAGES of the tile in the padded image;
// an image whose pre-image under the sample's map misses the resampled picture (the usual case for the reflections) is dropped by a bounding-box test.  Per image, every
// step the transpose of the forward kernel's step with the forward's own coefficients and tap arithmetic, in LDS:
//   0. (once per image, for all channels) resample^T as a GATHER: the tile's 16 x 16 padded-image pixels are fed by a (2 * 16 + 10)^2 block of up-sampled pixels; a thread
//      owns seven of them, finds the hi-res pixels whose bilinear footprint contains each (they lie in a 3 x 3 window around the pixel's pre-image for every map that
//      moves less than ~1.5 hi-res pixels per up-sampled pixel, a 4 x 4 window below ~2 -- those in 8 x 8 sub-tiles), evaluates their weights with the forward's coordinate arithmetic and keeps
//      the 9 / 16 weights (zeros for non-contributors) in registers;
//   a. per channel: the hi-res pixels those windows reach lie in the pre-image of the block's footprint: its bounding box (<= ADJ_GB^2) fixes the box of dy pixels
//      needed ((box / 2 + 6)^2), read once;
//   b. down2^T: zero-insertion + FIR, separable (one thread makes an odd and an even hi-res position from the same six dy values);
//   c. resample^T: 9 / 16 LDS reads + FMAs per up-sampled pixel with the kept weights -- fixed summation order, no atomics (a window word outside the box meets a
//      zero weight: the buffers start as zeros so that such a word is finite; a NaN / Inf in dy reaches the up-sampled pixels within a window of it rather than its
//      four taps only);
//   d. up2^T: 12-tap FIR + decimation, separable; the last step lands in dx (first image: store; further images: the same thread adds), mirrored images read their
//      row / column backwards.
// A sample whose box does not fit (zoom-in by more than ~1.25 at 45 degrees, 1.8 axis-aligned) is served in 8 x 8, 4 x 4 or 2 x 2 sub-tiles by the same code; one that
// moves 2 or more hi-res pixels per up-sampled pixel (zoom-in beyond 1.4 .. 2) takes step c as a scatter with LDS float atomics (measured 6x slower: 0.37 atomic
// lane-operations per clock and CU); one whose map is singular / not finite / zooms in by more than ~3.4 gets zeros here and the atomics kernel below
// (`..._rest_kernel`, which returns at once for every other sample).

constexpr int ADJ_T = 16;                     // dx tile
constexpr int ADJ_NT = 256;                   // threads
constexpr int ADJ_GB = 80;                    // largest staged box of hi-res gradient pixels (even)
constexpr int ADJ_DB = ADJ_GB / 2 + 7;        // dy box under it (odd pitch)
constexpr int ADJ_UB = 2 * ADJ_T + 10;        // up-sampled block over a full tile
constexpr int ADJ_NU = (ADJ_UB * ADJ_UB + ADJ_NT - 1) / ADJ_NT;      // up-sampled pixels a thread owns (full tile: 3 x 3 windows)
constexpr int ADJ_NU4 = (26 * 26 + ADJ_NT - 1) / ADJ_NT;              // ... in an 8 x 8 sub-tile (4 x 4 windows: 16 weights per pixel)

struct adj_map { float a, b, d, e, cx, cy, det, hx, hy; };

// affine map hi-res pixel -> up-sampled pixel in pixel units (ix = a X + b Y + cx, iy = d X + e Y + cy), and the tile size the gather form can serve (0: none)
__device__ __forceinline__ int adj_plan(const geom_params& p, const float* th, adj_map& m) {
    const float wu = 2.f * (p.w + p.mx0 + p.mx1), hu = 2.f * (p.h + p.my0 + p.my1), wo = 2.f * (p.w + 6), ho = 2.f * (p.h + 6);
    m.a = th[0] * wu / wo; m.b = th[1] * wu / ho;
    m.d = th[3] * hu / wo; m.e = th[4] * hu / ho;
    m.cx = ((th[0] * (1.f / wo - 1.f) + th[1] * (1.f / ho - 1.f) + th[2] + 1.f) * wu - 1.f) * 0.5f;
    m.cy = ((th[3] * (1.f / wo - 1.f) + th[4] * (1.f / ho - 1.f) + th[5] + 1.f) * hu - 1.f) * 0.5f;
    m.det = m.a * m.e - m.b * m.d;
    m.hx = m.hy = 0.f;
    if (!(fabsf(m.det) > 1e-12f) || !(fabsf(m.cx) < 1e6f) || !(fabsf(m.cy) < 1e6f)) return 0;
    m.hx = (fabsf(m.e) + fabsf(m.b)) / fabsf(m.det);     // half-widths of the bounding box of M^-1 [-1, 1]^2: hi-res pixels per up-sampled pixel
    m.hy = (fabsf(m.d) + fabsf(m.a)) / fabsf(m.det);
#pragma unroll
    for (int t = ADJ_T; t >= 2; t >>= 1) {
        const float r = (float)(2 * t + 11);      // the block's footprint rectangle: [u0 - 1, u0 + 2 t + 10]
        if (r * m.hx + 7.f <= (float)ADJ_GB && r * m.hy + 7.f <= (float)ADJ_GB) return t;    // (+ 7: floor / ceil, one pixel of slack per side, odd start, even length)
    }
    return 0;
}

struct adj_image { int active, qbx, dirx, pax, pbx, qlox, qby, diry, pay, pby, qloy, Xlo, Wb, Ylo, Hb; };

// one axis of one image: padded position of source pixel p is q = qb + dir * p for p in [pa, pb]; qlo = the smallest such q
__device__ __forceinline__ void adj_axis(int kind, int p0, int t, int size, int m0, int m1, int& qb, int& dir, int& pa, int& pb, int& qlo) {
    if (kind == 0) { qb = m0; dir = 1; pa = p0; pb = min(p0 + t, size) - 1; }
    else if (kind == 1) { qb = m0; dir = -1; pa = max(p0, 1); pb = min(min(p0 + t - 1, m0), size - 1); }
    else { qb = m0 + 2 * (size - 1); dir = -1; pa = max(p0, size - 1 - m1); pb = min(p0 + t - 1, size - 2); }
    qlo = dir > 0 ? qb + pa : qb - pb;
}

__device__ __forceinline__ int adj_clampi(float v) { return (int)fminf(fmaxf(v, -4e6f), 4e6f); }

// lanes per row of a (rows x cols) step: the smallest power of two >= cols (cols <= 128); an item index splits into (row, column) by shift / mask
__device__ __forceinline__ int adj_lg(int cols) { return cols <= 16 ? 4 : cols <= 32 ? 5 : cols <= 64 ? 6 : 7; }

__global__ __launch_bounds__(ADJ_NT, 3) void ada_geometric_adjoint_kernel(geom_params p) {      // p.x = dy, p.y = dx
    __shared__ __attribute__((aligned(16))) float s_a[ADJ_DB * ADJ_DB];                  // dy box; later the up-sampled gradient block [ADJ_UB][ADJ_UB + 1]
    __shared__ __attribute__((aligned(16))) float s_b[ADJ_DB * ADJ_GB];                  // dy rows expanded along x; later the vertically decimated block [t][ADJ_UB]
    __shared__ __attribute__((aligned(16))) float s_cs[(ADJ_GB + 6) * (ADJ_GB + 1) + 8];  // hi-res gradient box, three guard rows (+ 4 words) either side
    __shared__ adj_image s_img[9];
    static_assert(ADJ_UB * (ADJ_UB + 1) <= ADJ_DB * ADJ_DB && ADJ_T * ADJ_UB <= ADJ_DB * ADJ_GB, "aliased buffers");
    float* s_gu = s_a;
    float* s_v = s_b;
    float* s_c = s_cs + 3 * (ADJ_GB + 1) + 4;

    const int tid = threadIdx.x;
    const int n = blockIdx.z;
    const int wp = p.w + p.mx0 + p.mx1, hp = p.h + p.my0 + p.my1, wu = 2 * wp, hu = 2 * hp;
    const int wo = 2 * (p.w + 6), ho = 2 * (p.h + 6);
    const float* th = p.theta + (size_t)n * 6;
    const float t0 = th[0], t1 = th[1], t2 = th[2], t3 = th[3], t4 = th[4], t5 = th[5];
    const size_t plane_sz = (size_t)p.h * p.w;
    // the gather of step c reads words its zero weights cancel: none may hold a NaN / Inf pattern left by another kernel
    for (int e = threadIdx.x; e < ADJ_DB * ADJ_DB; e += ADJ_NT) s_a[e] = 0.f;
    for (int e = threadIdx.x; e < ADJ_DB * ADJ_GB; e += ADJ_NT) s_b[e] = 0.f;
    for (int e = threadIdx.x; e < (ADJ_GB + 6) * (ADJ_GB + 1) + 8; e += ADJ_NT) s_cs[e] = 0.f;
    adj_map m;
    int T = adj_plan(p, th, m);
    if (T == 0) {        // the atomics kernel adds this sample's gradient onto zeros
        const int py = blockIdx.y * ADJ_T + tid / ADJ_T, px = blockIdx.x * ADJ_T + tid % ADJ_T;
        if (tid < ADJ_T * ADJ_T && px < p.w && py < p.h)
            for (int ch = 0; ch < p.c; ch++) p.y[((size_t)n * p.c + ch) * plane_sz + (size_t)py * p.w + px] = 0.f;
        return;
    }
    // contributors of an up-sampled pixel fit a 3 x 3 / 4 x 4 window of hi-res pixels (an interval of length 2 h + 0.02 holds at most floor(2 h + 0.02) + 1 integers)
    const int win = (m.hx < 1.48f && m.hy < 1.48f) ? 3 : (m.hx < 1.98f && m.hy < 1.98f) ? 4 : 0;
    if (win == 4) T = min(T, 8);                            // (16 weights per pixel: fewer pixels per thread)
    const int UBT = 2 * T + 10;
    const int nsub = ADJ_T / T;
    const float inv_det = 1.f / m.det, inv_ubt = 1.f / (float)UBT;

    // affine_resample_kernel's arithmetic for one hi-res pixel
    auto source = [&](int X, int Y, float& ix, float& iy) {
        const float xn = (2 * X + 1) / (float)wo - 1.f, yn = (2 * Y + 1) / (float)ho - 1.f;
        const float gx = t0 * xn + t1 * yn + t2, gy = t3 * xn + t4 * yn + t5;
        ix = ((gx + 1.f) * wu - 1.f) * 0.5f;
        iy = ((gy + 1.f) * hu - 1.f) * 0.5f;
    };

    for (int sub = 0; sub < nsub * nsub; sub++) {
        const int px0 = blockIdx.x * ADJ_T + (sub % nsub) * T, py0 = blockIdx.y * ADJ_T + (sub / nsub) * T;
        if (px0 >= p.w || py0 >= p.h) continue;           // (uniform)
        __syncthreads();                                   // the previous sub-tile's readers of s_img
        if (tid < 9) {
            adj_image im;
            adj_axis(tid % 3, px0, T, p.w, p.mx0, p.mx1, im.qbx, im.dirx, im.pax, im.pbx, im.qlox);
            adj_axis(tid / 3, py0, T, p.h, p.my0, p.my1, im.qby, im.diry, im.pay, im.pby, im.qloy);
            im.active = im.pax <= im.pbx && im.pay <= im.pby;
            im.Xlo = im.Ylo = 1; im.Wb = im.Hb = 0;
            if (im.active) {
                // footprint rectangle of the up-sampled block (clipped to the up-sampled image + 1: nothing outside it receives gradient)
                const float rx0 = fmaxf((float)(2 * im.qlox - 6), -1.f), rx1 = fminf((float)(2 * im.qlox - 5 + UBT), (float)wu);
                const float ry0 = fmaxf((float)(2 * im.qloy - 6), -1.f), ry1 = fminf((float)(2 * im.qloy - 5 + UBT), (float)hu);
                float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float ux = ((k & 1) ? rx1 : rx0) - m.cx, uy = ((k & 2) ? ry1 : ry0) - m.cy;
                    const float X = (m.e * ux - m.b * uy) * inv_det, Y = (-m.d * ux + m.a * uy) * inv_det;
                    xmin = fminf(xmin, X); xmax = fmaxf(xmax, X); ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
                }
                int Xlo = max(adj_clampi(floorf(xmin)) - 1, 1), Xhi = min(adj_clampi(ceilf(xmax)) + 1, wo - 1);     // (hi-res column / row 0 is read by no output)
                int Ylo = max(adj_clampi(floorf(ymin)) - 1, 1), Yhi = min(adj_clampi(ceilf(ymax)) + 1, ho - 1);
                Xlo -= !(Xlo & 1); Ylo -= !(Ylo & 1);         // odd start: the pair (odd, even) shares its dy taps
                im.Xlo = Xlo; im.Ylo = Ylo;
                im.Wb = min((Xhi - Xlo + 2) & ~1, ADJ_GB);
                im.Hb = min((Yhi - Ylo + 2) & ~1, ADJ_GB);
                if (Xhi < Xlo || Yhi < Ylo || rx0 > rx1 || ry0 > ry1) im.active = 0;
            }
            s_img[tid] = im;
        }
        __syncthreads();

        const int lyp = tid / T, lxp = tid - lyp * T;       // this thread's pixel of the sub-tile (tid < T * T)
        const int px = px0 + lxp, py = py0 + lyp;
        const bool mine = tid < T * T && px < p.w && py < p.h;
        float* const out = p.y + (size_t)n * p.c * plane_sz + (size_t)py * p.w + px;

        if (!s_img[0].active && mine)                       // (the tile itself sees nothing of the picture: the reflections, if any, add onto zeros)
            for (int ch = 0; ch < p.c; ch++) out[(size_t)ch * plane_sz] = 0.f;

        // one image of the tile, all channels; WIN = 3 / 4: resample^T as a gather over WIN x WIN windows, 0: as a scatter with LDS atomics
        auto image_pass = [&](const adj_image& im, bool add, auto win_c, auto nu_c) {
            constexpr int WIN = decltype(win_c)::value, NUK = decltype(nu_c)::value;
            constexpr int NW = WIN ? WIN * WIN : 1;
            const int Xlo = im.Xlo, Ylo = im.Ylo, Wb = im.Wb, Hb = im.Hb, W2 = Wb >> 1, H2 = Hb >> 1;
            const int oxlo = ((Xlo - 1) >> 1) - 5, oylo = ((Ylo - 1) >> 1) - 5;
            const int dbw = W2 + 5, dbh = H2 + 5;
            const int U0x = 2 * im.qlox - 5, U0y = 2 * im.qloy - 5;
            const bool here = mine && px >= im.pax && px <= im.pbx && py >= im.pay && py <= im.pby;
            const int qxl = im.qbx + im.dirx * px - im.qlox, qyl = im.qby + im.diry * py - im.qloy;
            const int lg_b1 = adj_lg((W2 + 1) >> 1), lg_b2 = adj_lg(Wb);

            // 0. resample^T as a gather: this thread's up-sampled pixels, their candidate windows, the forward's weights (1 - |ix - ux|) (1 - |iy - uy|), zero
            //    for a hi-res pixel whose footprint misses the pixel
            constexpr int WN = WIN ? WIN : 1;
            float wt[NUK][NW];
            int wbase[NUK], woff[NUK];
            if (WIN) {
#pragma unroll
                for (int k = 0; k < NUK; k++) {
                    const int idx = tid + k * ADJ_NT;
                    const int uyl = (int)(((float)idx + 0.5f) * inv_ubt), uxl = idx - uyl * UBT;
                    woff[k] = idx < UBT * UBT ? uyl * (ADJ_UB + 1) + uxl : ADJ_DB * ADJ_DB - 1;      // (a word of s_a nothing reads)
                    const int ux = U0x + uxl, uy = U0y + uyl;
                    const bool inside = idx < UBT * UBT && ux >= 0 && ux < wu && uy >= 0 && uy < hu;
                    const float dux = (float)ux - m.cx, duy = (float)uy - m.cy;
                    const float qx = (m.e * dux - m.b * duy) * inv_det, qy = (-m.d * dux + m.a * duy) * inv_det;
                    const int X0 = adj_clampi(ceilf(qx - m.hx - 0.01f)), Y0 = adj_clampi(ceilf(qy - m.hy - 0.01f));
                    float xn[WN], yn[WN];
                    bool vx[WN], vy[WN];
#pragma unroll
                    for (int j = 0; j < WN; j++) {
                        const int X = X0 + j, Y = Y0 + j;
                        xn[j] = (2 * X + 1) / (float)wo - 1.f; yn[j] = (2 * Y + 1) / (float)ho - 1.f;      // (the forward's expressions)
                        vx[j] = X >= Xlo && X < Xlo + Wb && X < wo; vy[j] = inside && Y >= Ylo && Y < Ylo + Hb && Y < ho;
                    }
                    // a window with a column and a row inside the box starts at most WN - 1 rows / columns before it and ends at most WN - 1 behind it: inside the
                    // guard band of s_c; any other window holds zero weights only and reads the box's first words
                    bool anyx = false, anyy = false;
#pragma unroll
                    for (int j = 0; j < WN; j++) { anyx |= vx[j]; anyy |= vy[j]; }
                    wbase[k] = anyx && anyy ? (Y0 - Ylo) * (ADJ_GB + 1) + (X0 - Xlo) : 0;
#pragma unroll
                    for (int j = 0; j < NW; j++) {
                        const float xn_ = xn[j % WN], yn_ = yn[j / WN];
                        const float gx = t0 * xn_ + t1 * yn_ + t2, gy = t3 * xn_ + t4 * yn_ + t5;
                        const float ix = ((gx + 1.f) * wu - 1.f) * 0.5f, iy = ((gy + 1.f) * hu - 1.f) * 0.5f;
                        const float wx = fmaxf(1.f - fabsf(ix - (float)ux), 0.f), wy = fmaxf(1.f - fabsf(iy - (float)uy), 0.f);
                        wt[k][j] = vx[j % WN] && vy[j / WN] ? wx * wy : 0.f;
                    }
                }
            }

            const float inv_dbw = 1.f / (float)dbw;      // item -> (row, column) of the dy box by a float reciprocal ((item + 0.5) / dbw is never within 0.01 of an integer)
            const int nbox = dbh * dbw;

            for (int ch = 0; ch < p.c; ch++) {
                // a. dy box (zeros outside the picture)
                {
                    const float* plane = p.x + ((size_t)n * p.c + ch) * plane_sz;
                    for (int idx = tid; idx < nbox; idx += ADJ_NT) {
                        const int r = (int)(((float)idx + 0.5f) * inv_dbw), ci = idx - r * dbw;
                        const int oy = oylo + r, ox = oxlo + ci;
                        s_a[r * ADJ_DB + ci] = (ox >= 0 && ox < p.w && oy >= 0 && oy < p.h) ? plane[(unsigned)(oy * p.w + ox)] : 0.f;
                    }
                }
                __syncthreads();
                // b1. down2^T along x: hi-res columns Xlo + 2k (odd: taps fd[0], fd[2], ..) and Xlo + 2k + 1 (even: fd[1], fd[3], ..) from dy columns k .. k + 5;
                //     a thread makes k and k + 1 (four hi-res columns) from seven dy values
                for (int idx = tid; idx < (dbh << lg_b1); idx += ADJ_NT) {
                    const int r = idx >> lg_b1, k = (idx & ((1 << lg_b1) - 1)) * 2;
                    if (k >= W2) continue;
                    const float* q = s_a + r * ADJ_DB + k;
                    float v[7];
#pragma unroll
                    for (int j = 0; j < 7; j++) v[j] = q[j];
                    float o0 = 0.f, e0 = 0.f, o1 = 0.f, e1 = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        o0 = __builtin_fmaf(p.fd[2 * j], v[5 - j], o0); e0 = __builtin_fmaf(p.fd[2 * j + 1], v[5 - j], e0);
                        o1 = __builtin_fmaf(p.fd[2 * j], v[6 - j], o1); e1 = __builtin_fmaf(p.fd[2 * j + 1], v[6 - j], e1);
                    }
                    float* w_ = s_b + r * ADJ_GB + 2 * k;
                    if (k + 1 < W2) *reinterpret_cast<float4*>(w_) = make_float4(o0, e0, o1, e1);
                    else *reinterpret_cast<float2*>(w_) = make_float2(o0, e0);
                }
                __syncthreads();
                // b2. ... along y: a thread owns one column and a run of rows (wave w: the w-th quarter), two pairs per step -- rows 2k .. 2k + 3 from the seven
                //     expanded dy rows k .. k + 6, five of which the previous step already holds
                {
                    const int npp = (H2 + 1) >> 1, seg = (npp + 3) >> 2;
                    const int kk0 = (tid >> 6) * seg, kk1 = min(npp, kk0 + seg);
                    for (int j_ = tid & 63; j_ < Wb; j_ += 64) {
                        if (kk0 >= kk1) break;
                        const float* q = s_b + (2 * kk0) * ADJ_GB + j_;
                        float* w_ = s_c + (4 * kk0) * (ADJ_GB + 1) + j_;
                        float v[7];
#pragma unroll
                        for (int j = 0; j < 5; j++) v[j + 2] = q[j * ADJ_GB];
                        for (int kk = kk0; kk < kk1; kk++) {
#pragma unroll
                            for (int j = 0; j < 5; j++) v[j] = v[j + 2];
                            v[5] = q[5 * ADJ_GB]; v[6] = q[6 * ADJ_GB];
                            float o0 = 0.f, e0 = 0.f, o1 = 0.f, e1 = 0.f;
#pragma unroll
                            for (int j = 0; j < 6; j++) {
                                o0 = __builtin_fmaf(p.fd[2 * j], v[5 - j], o0); e0 = __builtin_fmaf(p.fd[2 * j + 1], v[5 - j], e0);
                                o1 = __builtin_fmaf(p.fd[2 * j], v[6 - j], o1); e1 = __builtin_fmaf(p.fd[2 * j + 1], v[6 - j], e1);
                            }
                            w_[0] = o0; w_[ADJ_GB + 1] = e0;
                            if (2 * kk + 1 < H2) { w_[2 * (ADJ_GB + 1)] = o1; w_[3 * (ADJ_GB + 1)] = e1; }
                            q += 2 * ADJ_GB; w_ += 4 * (ADJ_GB + 1);
                        }
                    }
                }
                if (!WIN)
                    for (int e = tid; e < UBT * (ADJ_UB + 1); e += ADJ_NT) s_gu[e] = 0.f;       // (the dy box is dead)
                __syncthreads();
                // c. resample^T
                if (WIN) {
#pragma unroll
                    for (int k = 0; k < NUK; k++) {
                        const float* q = s_c + wbase[k];
                        float v = 0.f;
#pragma unroll
                        for (int j = 0; j < NW; j++)      // (a word outside the box has weight zero: a stale but finite value -- the buffers start as zeros)
                            v = __builtin_fmaf(wt[k][j], q[(j / (WIN ? WIN : 1)) * (ADJ_GB + 1) + (j % (WIN ? WIN : 1))], v);
                        s_gu[woff[k]] = v;
                    }
                } else {
                    // the forward's taps, scattered (LDS float atomics: the order of the additions into one word is not fixed)
                    for (int idx = tid; idx < (Hb << lg_b2); idx += ADJ_NT) {
                        const int ly = idx >> lg_b2, lx = idx & ((1 << lg_b2) - 1);
                        const int X = Xlo + lx, Y = Ylo + ly;
                        if (lx >= Wb || X >= wo || Y >= ho) continue;
                        const float g = s_c[ly * (ADJ_GB + 1) + lx];
                        float ix, iy;
                        source(X, Y, ix, iy);
                        const float fx = floorf(ix), fy = floorf(iy);
                        const float tx = ix - fx, ty = iy - fy;
                        const int x0 = adj_clampi(fx), y0 = adj_clampi(fy);
                        const int bx = x0 - U0x, by = y0 - U0y;
                        const bool vx0 = x0 >= 0 && x0 < wu && bx >= 0 && bx < UBT, vx1 = x0 + 1 >= 0 && x0 + 1 < wu && bx + 1 >= 0 && bx + 1 < UBT;
                        const bool vy0 = y0 >= 0 && y0 < hu && by >= 0 && by < UBT, vy1 = y0 + 1 >= 0 && y0 + 1 < hu && by + 1 >= 0 && by + 1 < UBT;
                        float* q = s_gu + by * (ADJ_UB + 1) + bx;
                        if (vy0 && vx0) atomicAdd(q, (1.f - tx) * (1.f - ty) * g);
                        if (vy0 && vx1) atomicAdd(q + 1, tx * (1.f - ty) * g);
                        if (vy1 && vx0) atomicAdd(q + ADJ_UB + 1, (1.f - tx) * ty * g);
                        if (vy1 && vx1) atomicAdd(q + ADJ_UB + 2, tx * ty * g);
                    }
                }
                __syncthreads();
                // d1. up2^T along y: padded rows qloy + r, r + 1 <- up-sampled rows 2 r .. 2 r + 13 of the block (taps 2 f[k]: fe / fo interleaved); a thread owns one
                //     column, wave w the row pairs w, w + 4, ..
                if ((tid & 63) < UBT) {
                    const int j_ = tid & 63;
                    for (int rr = tid >> 6; rr < (T >> 1); rr += ADJ_NT / 64) {
                        const float* q = s_gu + (4 * rr) * (ADJ_UB + 1) + j_;
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            const float v0 = q[(2 * k) * (ADJ_UB + 1)], v1 = q[(2 * k + 1) * (ADJ_UB + 1)];
                            a0 = __builtin_fmaf(p.fo[5 - k], v0, a0); a0 = __builtin_fmaf(p.fe[5 - k], v1, a0);
                            if (k > 0) { a1 = __builtin_fmaf(p.fo[6 - k], v0, a1); a1 = __builtin_fmaf(p.fe[6 - k], v1, a1); }
                        }
                        a1 = __builtin_fmaf(p.fo[0], q[12 * (ADJ_UB + 1)], a1); a1 = __builtin_fmaf(p.fe[0], q[13 * (ADJ_UB + 1)], a1);
                        s_v[(2 * rr) * ADJ_UB + j_] = a0;
                        s_v[(2 * rr + 1) * ADJ_UB + j_] = a1;
                    }
                }
                __syncthreads();
                // d2. ... along x, into dx: the tile's own image stores, a reflection adds (the same thread, the same address)
                if (here) {
                    const float* q = s_v + qyl * ADJ_UB + 2 * qxl;
                    float v = 0.f;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        v = __builtin_fmaf(p.fo[5 - k], q[2 * k], v);
                        v = __builtin_fmaf(p.fe[5 - k], q[2 * k + 1], v);
                    }
                    float* o = out + (size_t)ch * plane_sz;
                    *o = add ? *o + v : v;
                }
                // (the next channel's step a writes s_a = s_gu, last read before the barrier above; its step b1 writes s_b = s_v behind its own barrier)
            }
        };

        for (int img = 0; img < 9; img++) {
            const adj_image im = s_img[img];
            if (!im.active) continue;                       // (uniform)
            if (win == 3) image_pass(im, img != 0, std::integral_constant<int, 3>{}, std::integral_constant<int, ADJ_NU>{});
            else if (win == 4) image_pass(im, img != 0, std::integral_constant<int, 4>{}, std::integral_constant<int, ADJ_NU4>{});
            else image_pass(im, img != 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        }
    }
}

// the atomics form for the samples the kernel above zero-filled: one thread per hi-res pixel, the whole chain scattered (4 x 36 atomics per pixel and channel)
__global__ __launch_bounds__(256) void ada_geometric_adjoint_rest_kernel(geom_params p) {
    const int n = blockIdx.z;
    const float* th = p.theta + (size_t)n * 6;
    adj_map m;
    if (adj_plan(p, th, m) != 0) return;
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int wp = p.w + p.mx0 + p.mx1, hp = p.h + p.my0 + p.my1, wu = 2 * wp, hu = 2 * hp;
    const int wo = 2 * (p.w + 6), ho = 2 * (p.h + 6);
    if (X < 1 || Y < 1 || X >= wo || Y >= ho) return;
    const float xn = (2 * X + 1) / (float)wo - 1.f, yn = (2 * Y + 1) / (float)ho - 1.f;
    const float gx = th[0] * xn + th[1] * yn + th[2], gy = th[3] * xn + th[4] * yn + th[5];
    const float lim = 1e7f;
    float ix = ((gx + 1.f) * wu - 1.f) * 0.5f, iy = ((gy + 1.f) * hu - 1.f) * 0.5f;
    if (!(ix == ix) || !(iy == iy)) return;                // (NaN map: the forward's taps fall outside)
    ix = fminf(fmaxf(ix, -lim), lim); iy = fminf(fmaxf(iy, -lim), lim);
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    const int x0 = (int)fx, y0 = (int)fy;
    const size_t plane_sz = (size_t)p.h * p.w;
    const int bx_ = (X - 1) >> 1, by_ = (Y - 1) >> 1, ex = !(X & 1), ey = !(Y & 1);
    for (int ch = 0; ch < p.c; ch++) {
        const float* plane = p.x + ((size_t)n * p.c + ch) * plane_sz;
        float* out = p.y + ((size_t)n * p.c + ch) * plane_sz;
        float g = 0.f;
        for (int jy = 0; jy < 6; jy++) {
            const int oy = by_ - jy;
            if (oy < 0 || oy >= p.h) continue;
            float row = 0.f;
            for (int jx = 0; jx < 6; jx++) {
                const int ox = bx_ - jx;
                if (ox >= 0 && ox < p.w) row = __builtin_fmaf(p.fd[2 * jx + ex], plane[(size_t)oy * p.w + ox], row);
            }
            g = __builtin_fmaf(p.fd[2 * jy + ey], row, g);
        }
        if (g == 0.f) continue;
        for (int tap = 0; tap < 4; tap++) {
            const int ux = x0 + (tap & 1), uy = y0 + (tap >> 1);
            if (ux < 0 || ux >= wu || uy < 0 || uy >= hu) continue;
            const float wgt = ((tap & 1) ? tx : 1.f - tx) * ((tap >> 1) ? ty : 1.f - ty) * g;
            // up-sampled pixel 2 i <- fe[t] P[i + t - 3];  2 i + 1 <- fo[t] P[i + t - 2]
            const int qx0 = (ux >> 1) - 3 + (ux & 1), qy0 = (uy >> 1) - 3 + (uy & 1);
            for (int r = 0; r < 6; r++) {
                const int qy = qy0 + r;
                if (qy < 0 || qy >= hp) continue;
                const float wr = ((uy & 1) ? p.fo[r] : p.fe[r]) * wgt;
                const int sy = geo_reflect(qy - p.my0, p.h);
                for (int t = 0; t < 6; t++) {
                    const int qx = qx0 + t;
                    if (qx < 0 || qx >= wp) continue;
                    atomicAdd(out + (size_t)sy * p.w + geo_reflect(qx - p.mx0, p.w), ((ux & 1) ? p.fo[t] : p.fe[t]) * wr);
                }
            }
        }
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

extern "C" int sgv_ada_geometric(const float* x, float* y, const float* theta, const float* filter12, int32_t n, int32_t c, int32_t h, int32_t w,
                                 int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, void* stream_) {
    if (!x || !y || !theta || !filter12) return sgv_fail(SGV_ERR_INVALID_ARG, "ada_geometric: NULL pointer");
    if (n < 1 || c < 1 || h < 2 || w < 2) return sgv_fail(SGV_ERR_INVALID_ARG, "ada_geometric: sizes must be positive (images of at least 2x2)");
    if (mx0 < 0 || mx1 < 0 || my0 < 0 || my1 < 0 || mx0 > w - 1 || mx1 > w - 1 || my0 > h - 1 || my1 > h - 1)
        return sgv_fail(SGV_ERR_INVALID_ARG, "ada_geometric: the reflect margin must lie in [0, size - 1]");
    if (n > 65535 || (h + GEO_TO - 1) / GEO_TO > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "ada_geometric: batch / height too large");
    if ((int64_t)n * c * h * w > INT32_MAX || (int64_t)3 * w > (1 << 24) || (int64_t)3 * h > (1 << 24)) return sgv_fail(SGV_ERR_TOO_LARGE, "ada_geometric: tensors are too large");
    geom_params p{};
    p.x = x; p.y = y; p.theta = theta;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.mx0 = mx0; p.mx1 = mx1; p.my0 = my0; p.my1 = my1;
    // host-readable filter: 12 taps (the caller passes a host pointer: the taps are launch arguments)
    for (int t = 0; t < 6; t++) { p.fe[t] = 2.f * filter12[11 - 2 * t]; p.fo[t] = 2.f * filter12[10 - 2 * t]; }
    for (int m = 0; m < 12; m++) p.fd[m] = filter12[m];
    hipStream_t stream = (hipStream_t)stream_;
    sgv_launch_scope scope(SGV_K_POINTWISE, stream, 8.0 * n * c * (double)h * w);
    dim3 grid((unsigned)((w + GEO_TO - 1) / GEO_TO), (unsigned)((h + GEO_TO - 1) / GEO_TO), (unsigned)n);
    hipLaunchKernelGGL(ada_geometric_forward_kernel, grid, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(ada_geometric_forward_sub_kernel, grid, dim3(256), 0, stream, p);      // (samples whose map needs sub-tiles: none at p = 0)
    return sgv_check_launch("ada_geometric_forward_kernel");
}

extern "C" int sgv_ada_geometric_adjoint(const float* dy, float* dx, const float* theta, const float* filter12, int32_t n, int32_t c, int32_t h, int32_t w,
                                         int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, void* stream_) {
    if (!dy || !dx || !theta || !filter12) return sgv_fail(SGV_ERR_INVALID_ARG, "ada_geometric_adjoint: NULL pointer");
    if (n < 1 || c < 1 || h < 2 || w < 2) return sgv_fail(SGV_ERR_INVALID_ARG, "ada_geometric_adjoint: sizes must be positive (images of at least 2x2)");
    if (mx0 < 0 || mx1 < 0 || my0 < 0 || my1 < 0 || mx0 > w - 1 || mx1 > w - 1 || my0 > h - 1 || my1 > h - 1)
        return sgv_fail(SGV_ERR_INVALID_ARG, "ada_geometric_adjoint: the reflect margin must lie in [0, size - 1]");
    if (n > 65535 || (2 * (h + 6) + 3) / 4 > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "ada_geometric_adjoint: batch / height too large");
    if ((int64_t)n * c * h * w > INT32_MAX || (int64_t)3 * w > (1 << 20) || (int64_t)3 * h > (1 << 20)) return sgv_fail(SGV_ERR_TOO_LARGE, "ada_geometric_adjoint: tensors are too large");
    geom_params p{};
    p.x = dy; p.y = dx; p.theta = theta;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.mx0 = mx0; p.mx1 = mx1; p.my0 = my0; p.my1 = my1;
    for (int t = 0; t < 6; t++) { p.fe[t] = 2.f * filter12[11 - 2 * t]; p.fo[t] = 2.f * filter12[10 - 2 * t]; }      // the forward's coefficients, bit for bit
    for (int m = 0; m < 12; m++) p.fd[m] = filter12[m];
    hipStream_t stream = (hipStream_t)stream_;
    sgv_launch_scope scope(SGV_K_POINTWISE, stream, 8.0 * n * c * (double)h * w);
    dim3 grid((unsigned)((w + ADJ_T - 1) / ADJ_T), (unsigned)((h + ADJ_T - 1) / ADJ_T), (unsigned)n);
    hipLaunchKernelGGL(ada_geometric_adjoint_kernel, grid, dim3(ADJ_NT), 0, stream, p);
    // samples the gather form cannot serve (singular / extreme maps): zero-filled above, scattered here; every other sample's workgroups return at once
    dim3 rgrid((unsigned)((2 * (w + 6) + 63) / 64), (unsigned)((2 * (h + 6) + 3) / 4), (unsigned)n);
    hipLaunchKernelGGL(ada_geometric_adjoint_rest_kernel, rgrid, dim3(256), 0, stream, p);
    return sgv_check_launch("ada_geometric_adjoint_kernel");
}
