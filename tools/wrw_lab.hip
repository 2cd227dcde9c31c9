// Lab for the 3x3 weight-gradient kernel (stylegan-v_amd/csrc/wrw_kernel.h): correctness against a naive fp32 kernel + fp64 host
// check on a small case, then timing on the training shapes.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Istylegan-v_amd/csrc tools/wrw_lab.hip -o tools/wrw_lab && tools/wrw_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "wrw_kernel.h"
#include "wrw_ws_kernel.h"
#include "wrw_s2_ws_kernel.h"

using namespace sgv_wrw;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; p[i] = ((h & 0xffffff) / 16777216.f - 0.5f) * 2.f; }
}

// one workgroup per (o, i): fp64 accumulation of all nine taps
__global__ void naive_wrw(const float* dy, const float* x, double* dw, int n, int o_, int i_, int h, int w) {
    const int o = blockIdx.x / i_, i = blockIdx.x % i_;
    double s[9] = {0};
    for (size_t q = threadIdx.x; q < (size_t)n * h * w; q += blockDim.x) {
        const int xx = q % w, yy = (q / w) % h, nn = q / ((size_t)w * h);
        const double d = dy[(((size_t)nn * o_ + o) * h + yy) * w + xx];
        for (int ky = 0; ky < 3; ky++) for (int kx = 0; kx < 3; kx++) {
            const int sy = yy + ky - 1, sx = xx + kx - 1;
            if (sy >= 0 && sy < h && sx >= 0 && sx < w) s[ky * 3 + kx] += d * x[(((size_t)nn * i_ + i) * h + sy) * w + sx];
        }
    }
    __shared__ double red[256];
    for (int k = 0; k < 9; k++) {
        red[threadIdx.x] = s[k]; __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
        if (threadIdx.x == 0) dw[(size_t)blockIdx.x * 9 + k] = red[0];
        __syncthreads();
    }
}

static wrw_params make(const float* dy, const float* x, float* dw, int n, int o, int i, int h, int w, int rows, int wgs) {
    wrw_params p{};
    p.dy = dy; p.x = x; p.dw = dw; p.n = n; p.o = o; p.i = i; p.h = h; p.w = w;
    p.rows = rows < h ? rows : h;
    p.tiles_i = i / TI;
    p.units = n * (w / SEG) * (h / p.rows);
    const int tiles = (o / TO) * (i / TI);
    int splits = wgs / tiles; if (splits < 1) splits = 1; if (splits > p.units) splits = p.units;
    p.splits = splits;
    p.scatter_flush = getenv("WRW_SCATTER") ? 1 : 0;   // the element-per-lane flush (A/B against flush_tile)
    return p;
}

template <int TERMS> static void launch(const wrw_params& p) {
    dim3 grid((p.o / TO) * p.tiles_i, p.splits);
    hipLaunchKernelGGL(wrw3x3_kernel<TERMS>, grid, dim3(256), 0, 0, p);
}

template <int TERMS, int VIEWS, int ABL> static void launch_ws_abl(const wrw_params& p) {
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<TERMS, VIEWS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(VIEWS))); attr = true; }
    dim3 grid((p.o / TO) * p.tiles_i, p.splits);
    hipLaunchKernelGGL((wrw3x3_ws_kernel<TERMS, VIEWS, ABL>), grid, dim3(512), wrw_ws_lds_bytes(VIEWS), 0, p);
}
template <int TERMS, int VIEWS> static void launch_ws(const wrw_params& p) {   // producer / consumer form (wrw_ws_kernel.h); WRW_ABL = 6 / 7: ablations
    static const int abl = getenv("WRW_ABL") ? atoi(getenv("WRW_ABL")) : 0;
    if (abl == 6) launch_ws_abl<TERMS, VIEWS, 6>(p); else if (abl == 7) launch_ws_abl<TERMS, VIEWS, 7>(p); else if (abl == 8) launch_ws_abl<TERMS, VIEWS, 8>(p); else if (abl == 10) launch_ws_abl<TERMS, VIEWS, 10>(p); else if (abl == 11) launch_ws_abl<TERMS, VIEWS, 11>(p); else if (abl == 12) launch_ws_abl<TERMS, VIEWS, 12>(p); else if (abl == 13) launch_ws_abl<TERMS, VIEWS, 13>(p); else if (abl == 14) launch_ws_abl<TERMS, VIEWS, 14>(p); else launch_ws_abl<TERMS, VIEWS, 0>(p);
}

__global__ void naive_wrw_s2(const float* sm, const float* big, double* dw, int n, int cs, int cb, int h, int w) {
    const int o = blockIdx.x / cb, i = blockIdx.x % cb;
    const int hb = 2 * h + 1, wb = 2 * w + 1;
    double s[9] = {0};
    for (size_t q = threadIdx.x; q < (size_t)n * h * w; q += blockDim.x) {
        const int xx = q % w, yy = (q / w) % h, nn = q / ((size_t)w * h);
        const double d = sm[(((size_t)nn * cs + o) * h + yy) * w + xx];
        for (int ky = 0; ky < 3; ky++) for (int kx = 0; kx < 3; kx++) s[ky * 3 + kx] += d * big[(((size_t)nn * cb + i) * hb + 2 * yy + ky) * wb + 2 * xx + kx];
    }
    __shared__ double red[256];
    for (int k = 0; k < 9; k++) {
        red[threadIdx.x] = s[k]; __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
        if (threadIdx.x == 0) dw[(size_t)blockIdx.x * 9 + k] = red[0];
        __syncthreads();
    }
}

template <int TERMS> static wrw_s2_params launch_s2(const float* sm, const float* big, float* dw, int n, int cs, int cb, int h, int w, int wgs) {
    wrw_s2_params p{};
    p.small = sm; p.big = big; p.dw = dw; p.n = n; p.cs = cs; p.cb = cb; p.h = h; p.w = w;
    p.rows = h < 32 ? h : 32;
    p.scatter_flush = getenv("WRW_SCATTER") ? 1 : 0;
    p.tiles_b = cb / TI;
    p.units = n * (w / SEG) * (h / p.rows);
    const int tiles = (cs / TO) * p.tiles_b;
    int splits = wgs / tiles; if (splits < 1) splits = 1; if (splits > p.units) splits = p.units;
    p.splits = splits;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)wrw3x3_s2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_LDS_BYTES));
                 CK(hipFuncSetAttribute((const void*)wrw3x3_s2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_LDS_BYTES)); attr = true; }
    static const int ws = getenv("WRW_S2_WS") ? atoi(getenv("WRW_S2_WS")) : 1;     // 0: 4-wave kernel; 1: producer / consumer; 6 / 7: its ablations
    if (ws) {
        static bool attr2 = false;
        if (!attr2) { CK(hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<TERMS, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES));
                      CK(hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<TERMS, false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES));
                      CK(hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<TERMS, false, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES));
                      CK(hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<TERMS, false, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES)); attr2 = true; }
        if (ws == 9) { hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<TERMS, false, 9>), dim3(tiles, p.splits), dim3(512), WRW_S2_WS_LDS_BYTES, 0, p); return p; }
        if (ws == 6) hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<TERMS, false, 6>), dim3(tiles, p.splits), dim3(512), WRW_S2_WS_LDS_BYTES, 0, p);
        else if (ws == 7) hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<TERMS, false, 7>), dim3(tiles, p.splits), dim3(512), WRW_S2_WS_LDS_BYTES, 0, p);
        else hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<TERMS, false, 0>), dim3(tiles, p.splits), dim3(512), WRW_S2_WS_LDS_BYTES, 0, p);
        return p;
    }
    hipLaunchKernelGGL(wrw3x3_s2_kernel<TERMS>, dim3(tiles, p.splits), dim3(256), WRW_S2_LDS_BYTES, 0, p);
    return p;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    if (argc > 2) {   // ---- stride-2 member only ----
        {
            const int n = 3, cs = 128, cb = 64, h = 16, w = 64, hb = 2 * h + 1, wb = 2 * w + 1;
            const size_t ns = (size_t)n * cs * h * w, nb = (size_t)n * cb * hb * wb, ndw = (size_t)cs * cb * 9;
            float *sm, *big, *dw; double* ref;
            CK(hipMalloc(&sm, ns * 4)); CK(hipMalloc(&big, nb * 4)); CK(hipMalloc(&dw, ndw * 4)); CK(hipMalloc(&ref, ndw * 8));
            fill<<<(ns + 255) / 256, 256>>>(sm, ns, 11u); fill<<<(nb + 255) / 256, 256>>>(big, nb, 23u);
            naive_wrw_s2<<<cs * cb, 256>>>(sm, big, ref, n, cs, cb, h, w);
            std::vector<double> r(ndw); std::vector<float> gpu(ndw);
            CK(hipMemcpy(r.data(), ref, ndw * 8, hipMemcpyDeviceToHost));
            for (int terms = 1; terms <= 3; terms += 2) for (int wgs : {256, 4}) {
                CK(hipMemset(dw, 0, ndw * 4));
                wrw_s2_params p = terms == 1 ? launch_s2<1>(sm, big, dw, n, cs, cb, h, w, wgs) : launch_s2<3>(sm, big, dw, n, cs, cb, h, w, wgs);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(gpu.data(), dw, ndw * 4, hipMemcpyDeviceToHost));
                double maxerr = 0, maxref = 0, sq = 0, sqr = 0; size_t worst = 0;
                for (size_t k = 0; k < ndw; k++) { double e = fabs(gpu[k] - r[k]); if (!(e <= maxerr)) { maxerr = e; worst = k; } if (fabs(r[k]) > maxref) maxref = fabs(r[k]); sq += e * e; sqr += r[k] * r[k]; }
                printf("check s2 terms=%d splits=%d: max abs err %.3e (max |ref| %.3e, rel-L2 %.3e) worst at cs=%zu cb=%zu tap=%zu gpu=%f ref=%f\n", terms, p.splits, maxerr, maxref,
                       sqrt(sq / sqr), worst / 9 / cb, (worst / 9) % cb, worst % 9, gpu[worst], r[worst]);
            }
            CK(hipFree(sm)); CK(hipFree(big)); CK(hipFree(dw)); CK(hipFree(ref));
        }
        struct { const char* name; int n, cb, cs, r; } sh[] = { {"big 64ch 257^2 | small 128ch 128^2", 96, 64, 128, 128}, {"big 128ch 129^2 | small 256ch 64^2", 96, 128, 256, 64}, {"big 256ch 65^2 | small 512ch 32^2", 96, 256, 512, 32} };
        for (auto& s : sh) {
            const int hb = 2 * s.r + 1;
            const size_t ns = (size_t)s.n * s.cs * s.r * s.r, nb = (size_t)s.n * s.cb * hb * hb, ndw = (size_t)s.cs * s.cb * 9;
            float *sm, *big, *dw;
            CK(hipMalloc(&sm, ns * 4)); CK(hipMalloc(&big, nb * 4)); CK(hipMalloc(&dw, ndw * 4));
            fill<<<(ns + 255) / 256, 256>>>(sm, ns, 5u); fill<<<(nb + 255) / 256, 256>>>(big, nb, 7u);
            const double flops = 2.0 * s.n * s.r * s.r * (double)s.cs * s.cb * 9;
            for (int terms = 1; terms <= 3; terms += 2) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                if (terms == 1) launch_s2<1>(sm, big, dw, s.n, s.cs, s.cb, s.r, s.r, 256); else launch_s2<3>(sm, big, dw, s.n, s.cs, s.cb, s.r, s.r, 256);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; r++) { if (terms == 1) launch_s2<1>(sm, big, dw, s.n, s.cs, s.cb, s.r, s.r, 256); else launch_s2<3>(sm, big, dw, s.n, s.cs, s.cb, s.r, s.r, 256); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                printf("%-38s terms=%d  %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)  %6.1f GB/s of input\n", s.name, terms, ms, flops / ms / 1e9, (ns + nb) * 4.0 / ms / 1e6);
            }
            CK(hipFree(sm)); CK(hipFree(big)); CK(hipFree(dw));
        }
        return 0;
    }
    // ---- correctness ----
    {
        const int n = 3, o = 64, i = 128, h = 64, w = 64;
        const size_t ndy = (size_t)n * o * h * w, nx = (size_t)n * i * h * w, ndw = (size_t)o * i * 9;
        float *dy, *x, *dw; double* ref;
        CK(hipMalloc(&dy, ndy * 4)); CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&dw, ndw * 4)); CK(hipMalloc(&ref, ndw * 8));
        fill<<<(ndy + 255) / 256, 256>>>(dy, ndy, 11u); fill<<<(nx + 255) / 256, 256>>>(x, nx, 23u);
        naive_wrw<<<o * i, 256>>>(dy, x, ref, n, o, i, h, w);
        std::vector<double> r(ndw); std::vector<float> gpu(ndw);
        CK(hipMemcpy(r.data(), ref, ndw * 8, hipMemcpyDeviceToHost));
        for (int terms = 1; terms <= 3; terms += 2) for (int rows : {64, 32, 16}) {
            CK(hipMemset(dw, 0, ndw * 4));
            wrw_params p = make(dy, x, dw, n, o, i, h, w, rows, 8);
            if (terms == 1) launch<1>(p); else launch<3>(p);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(gpu.data(), dw, ndw * 4, hipMemcpyDeviceToHost));
            double maxerr = 0, maxref = 0, sq = 0, sqr = 0; size_t worst = 0;
            for (size_t k = 0; k < ndw; k++) { double e = fabs(gpu[k] - r[k]); if (e > maxerr) { maxerr = e; worst = k; } if (fabs(r[k]) > maxref) maxref = fabs(r[k]); sq += e * e; sqr += r[k] * r[k]; }
            printf("check terms=%d rows=%d splits=%d: max abs err %.3e (max |ref| %.3e, rel-L2 %.3e) worst at o=%zu i=%zu tap=%zu gpu=%f ref=%f\n", terms, rows, p.splits,
                   maxerr, maxref, sqrt(sq / sqr), worst / 9 / i, (worst / 9) % i, worst % 9, gpu[worst], r[worst]);
        }
        for (int views : {1, 3}) for (int terms = 1; terms <= 3; terms += 2) for (int rows : {64, 32, 16}) for (int wgs : {8, 256}) {
            CK(hipMemset(dw, 0, ndw * 4));
            wrw_params p = make(dy, x, dw, n, o, i, h, w, rows, wgs);
            if (views == 1) { if (terms == 1) launch_ws<1, 1>(p); else launch_ws<3, 1>(p); } else { if (terms == 1) launch_ws<1, 3>(p); else launch_ws<3, 3>(p); }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(gpu.data(), dw, ndw * 4, hipMemcpyDeviceToHost));
            double maxerr = 0, maxref = 0, sq = 0, sqr = 0; size_t worst = 0;
            for (size_t k = 0; k < ndw; k++) { double e = fabs(gpu[k] - r[k]); if (!(e <= maxerr)) { maxerr = e; worst = k; } if (fabs(r[k]) > maxref) maxref = fabs(r[k]); sq += e * e; sqr += r[k] * r[k]; }
            printf("check ws views=%d terms=%d rows=%d splits=%d: max abs err %.3e (max |ref| %.3e, rel-L2 %.3e) worst at o=%zu i=%zu tap=%zu gpu=%f ref=%f\n", views, terms, rows, p.splits,
                   maxerr, maxref, sqrt(sq / sqr), worst / 9 / i, (worst / 9) % i, worst % 9, gpu[worst], r[worst]);
        }
        CK(hipFree(dy)); CK(hipFree(x)); CK(hipFree(dw)); CK(hipFree(ref));
    }
    {   // small heights: units of 1, 2, 3, 8 rows (the ring / dy-buffer schedule at its edges)
        for (int h : {1, 2, 3, 8}) {
            const int n = 5, o = 64, i = 64, w = 32;
            const size_t ndy = (size_t)n * o * h * w, nx = (size_t)n * i * h * w, ndw = (size_t)o * i * 9;
            float *dy, *x, *dw; double* ref;
            CK(hipMalloc(&dy, ndy * 4)); CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&dw, ndw * 4)); CK(hipMalloc(&ref, ndw * 8));
            fill<<<(ndy + 255) / 256, 256>>>(dy, ndy, 11u); fill<<<(nx + 255) / 256, 256>>>(x, nx, 23u);
            naive_wrw<<<o * i, 256>>>(dy, x, ref, n, o, i, h, w);
            std::vector<double> r(ndw); std::vector<float> gpu(ndw);
            CK(hipMemcpy(r.data(), ref, ndw * 8, hipMemcpyDeviceToHost));
            for (int wgs : {1, 2}) {
                CK(hipMemset(dw, 0, ndw * 4));
                wrw_params p = make(dy, x, dw, n, o, i, h, w, 32, wgs);
                launch_ws<3, 1>(p);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(gpu.data(), dw, ndw * 4, hipMemcpyDeviceToHost));
                double maxerr = 0, maxref = 0;
                for (size_t k = 0; k < ndw; k++) { double e = fabs(gpu[k] - r[k]); if (!(e <= maxerr)) maxerr = e; if (fabs(r[k]) > maxref) maxref = fabs(r[k]); }
                printf("check ws h=%d splits=%d: max abs err %.3e (max |ref| %.3e)\n", h, p.splits, maxerr, maxref);
            }
            CK(hipFree(dy)); CK(hipFree(x)); CK(hipFree(dw)); CK(hipFree(ref));
        }
    }
    // ---- timing ----
    struct { const char* name; int n, c, r; } shapes[] = { {"64ch 256^2", 96, 64, 256}, {"128ch 128^2", 96, 128, 128}, {"256ch 64^2", 96, 256, 64}, {"512ch 32^2", 96, 512, 32} };
    for (auto& s : shapes) {
        const size_t na = (size_t)s.n * s.c * s.r * s.r, ndw = (size_t)s.c * s.c * 9;
        float *dy, *x, *dw;
        CK(hipMalloc(&dy, na * 4)); CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&dw, ndw * 4));
        fill<<<(na + 255) / 256, 256>>>(dy, na, 5u); fill<<<(na + 255) / 256, 256>>>(x, na, 7u);
        const double flops = 2.0 * s.n * s.r * s.r * (double)s.c * s.c * 9;
        for (int views : {1, 3}) for (int rows : {32, 64}) {   // producer / consumer form, one workgroup per CU
            if (rows > s.r) continue;
            wrw_params p = make(dy, x, dw, s.n, s.c, s.c, s.r, s.r, rows, 256);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            if (views == 1) launch_ws<3, 1>(p); else launch_ws<3, 3>(p);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) { if (views == 1) launch_ws<3, 1>(p); else launch_ws<3, 3>(p); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("%-12s ws views=%d terms=3 rows=%2d grid=%3dx%-3d units/wg=%5.1f  %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)\n", s.name, views, p.rows, (s.c / TO) * p.tiles_i, p.splits,
                   (double)p.units / p.splits, ms, flops / ms / 1e9);
            fflush(stdout);
        }
        for (int terms = 3; terms <= 3; terms += 2) for (int rows : {32}) for (int wgs : {256}) {
            if (rows > s.r) continue;
            wrw_params p = make(dy, x, dw, s.n, s.c, s.c, s.r, s.r, rows, wgs);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipMemsetAsync(dw, 0, ndw * 4)); if (terms == 1) launch<1>(p); else launch<3>(p);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) { if (terms == 1) launch<1>(p); else launch<3>(p); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("%-12s terms=%d rows=%2d grid=%3dx%-3d units/wg=%5.1f  %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)  %6.1f GB/s of input\n", s.name, terms, p.rows,
                   (s.c / TO) * p.tiles_i, p.splits, (double)p.units / p.splits, ms, flops / ms / 1e9, 2.0 * na * 4 / ms / 1e6);
            fflush(stdout);
        }
        CK(hipFree(dy)); CK(hipFree(x)); CK(hipFree(dw));
    }
    return 0;
}
