// Lab for the 3x3 forward convolution kernel (stylegan-v_amd/csrc/conv3x3_kernel.h): check against a naive fp64 kernel, then time.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Istylegan-v_amd/csrc tools/conv_lab.hip -o tools/conv_lab && tools/conv_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "conv3x3_kernel.h"
#include "conv3x3_ws_kernel.h"

using namespace sgv_conv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; p[i] = ((h & 0xffffff) / 16777216.f - 0.5f) * 2.f * scale; }
}

__global__ void naive_conv(const float* x, const float* w, double* y, int n, int k, int m, int h, int wd, int mode) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * m * h * wd) return;
    const int X = idx % wd, Y = (idx / wd) % h, mm = (idx / ((size_t)wd * h)) % m, nn = idx / ((size_t)wd * h * m);
    double s = 0;
    for (int kk = 0; kk < k; kk++) for (int ky = 0; ky < 3; ky++) for (int kx = 0; kx < 3; kx++) {
        const int sy = Y + ky - 1, sx = X + kx - 1;
        if (sy < 0 || sy >= h || sx < 0 || sx >= wd) continue;
        const float wv = mode == 0 ? w[(((size_t)mm * k + kk) * 3 + ky) * 3 + kx] : w[(((size_t)kk * m + mm) * 3 + (2 - ky)) * 3 + (2 - kx)];
        s += (double)wv * x[(((size_t)nn * k + kk) * h + sy) * wd + sx];
    }
    y[idx] = s;
}

template <int TERMS> static void launch(const float* x, const float* w, float* y, u32x4* wprep, int n, int k, int m, int h, int wd, int mode, int grid) {
    const int total = (m / TM) * (k / KC) * 9 * 2 * TM;
    hipLaunchKernelGGL(conv3x3_prep_weights, dim3((total + 255) / 256), dim3(256), 0, 0, w, wprep, m, k, mode, TERMS);
    conv_params p{};
    p.x = x; p.wprep = wprep; p.y = y; p.n = n; p.k = k; p.m = m; p.h = h; p.w = wd;
    p.tiles = n * (h / TROWS) * (wd / SEG) * (m / TM);
    p.grid = grid < p.tiles ? grid : p.tiles;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)conv3x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                 CK(hipFuncSetAttribute((const void*)conv3x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); attr = true; }
    hipLaunchKernelGGL(conv3x3_kernel<TERMS>, dim3(p.grid), dim3(256), LDS_BYTES, 0, p);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const bool legacy = argc <= 2;   // `conv_lab 5 ws`: only the producer / consumer sections
    const bool ablate = argc > 3;    // `conv_lab 5 ws abl`: also the timing-only ablation variants of the producer / consumer kernel
    {   // ---- correctness ----
        for (int mode = 0; mode < 2; mode++) {
            const int n = 2, k = 32, m = 128, h = 32, wd = 64;
            const size_t nx = (size_t)n * k * h * wd, ny = (size_t)n * m * h * wd, nw = (size_t)m * k * 9;
            float *x, *w, *y; double* ref; u32x4* wprep;
            CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&ref, ny * 8)); CK(hipMalloc(&wprep, nw * 4));
            fill<<<(nx + 255) / 256, 256>>>(x, nx, 11u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 23u, 0.1f);
            naive_conv<<<(ny + 255) / 256, 256>>>(x, w, ref, n, k, m, h, wd, mode);
            std::vector<double> r(ny); std::vector<float> gpu(ny);
            CK(hipMemcpy(r.data(), ref, ny * 8, hipMemcpyDeviceToHost));
            for (int terms = 1; terms <= 3; terms += 2) for (int grid : {256, 3}) {
                CK(hipMemset(y, 0xff, ny * 4));
                if (terms == 1) launch<1>(x, w, y, wprep, n, k, m, h, wd, mode, grid); else launch<3>(x, w, y, wprep, n, k, m, h, wd, mode, grid);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(gpu.data(), y, ny * 4, hipMemcpyDeviceToHost));
                double maxerr = 0, maxref = 0, sq = 0, sqr = 0; size_t worst = 0;
                for (size_t q = 0; q < ny; q++) { double e = fabs(gpu[q] - r[q]); if (!(e <= maxerr)) { maxerr = e; worst = q; } if (fabs(r[q]) > maxref) maxref = fabs(r[q]); sq += e * e; sqr += r[q] * r[q]; }
                printf("check mode=%d terms=%d grid=%d: max abs err %.3e (max |ref| %.3e, rel-L2 %.3e) worst idx %zu gpu=%f ref=%f\n", mode, terms, grid, maxerr, maxref, sqrt(sq / sqr), worst, gpu[worst], r[worst]);
            }
            CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(ref)); CK(hipFree(wprep));
        }
    }
    if (legacy) for (int sw : {16, 8}) for (int mode = 0; mode < 2; mode++) {   // ---- small-image kernel: correctness + timing ----
        const int n = 16, k = 32, m = 128;
        const size_t nx = (size_t)n * k * sw * sw, ny = (size_t)n * m * sw * sw, nw = (size_t)m * k * 9;
        float *x, *w, *y; double* ref; u32x4* wprep;
        CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&ref, ny * 8)); CK(hipMalloc(&wprep, nw * 4));
        fill<<<(nx + 255) / 256, 256>>>(x, nx, 11u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 23u, 0.1f);
        naive_conv<<<(ny + 255) / 256, 256>>>(x, w, ref, n, k, m, sw, sw, mode);
        std::vector<double> r(ny); std::vector<float> gpu(ny);
        CK(hipMemcpy(r.data(), ref, ny * 8, hipMemcpyDeviceToHost));
        const int total = (m / TM) * (k / KC) * 9 * 2 * TM;
        hipLaunchKernelGGL(conv3x3_prep_weights, dim3((total + 255) / 256), dim3(256), 0, 0, w, wprep, m, k, mode, 3);
        conv_params p{};
        p.x = x; p.wprep = wprep; p.y = y; p.n = n; p.k = k; p.m = m; p.h = sw; p.w = sw;
        for (int grid : {256, 3}) {
            CK(hipMemset(y, 0xff, ny * 4));
            if (sw == 16) { p.tiles = (n / small_cfg<16>::S) * (m / TM); p.grid = grid < p.tiles ? grid : p.tiles;
                CK(hipFuncSetAttribute((const void*)conv3x3_small_kernel<3, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<16>::LDS));
                hipLaunchKernelGGL((conv3x3_small_kernel<3, 16>), dim3(p.grid), dim3(256), small_cfg<16>::LDS, 0, p); }
            else { p.tiles = (n / small_cfg<8>::S) * (m / TM); p.grid = grid < p.tiles ? grid : p.tiles;
                CK(hipFuncSetAttribute((const void*)conv3x3_small_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<8>::LDS));
                hipLaunchKernelGGL((conv3x3_small_kernel<3, 8>), dim3(p.grid), dim3(256), small_cfg<8>::LDS, 0, p); }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(gpu.data(), y, ny * 4, hipMemcpyDeviceToHost));
            double maxerr = 0, maxref = 0, sq = 0, sqr = 0; size_t worst = 0;
            for (size_t q = 0; q < ny; q++) { double e = fabs(gpu[q] - r[q]); if (!(e <= maxerr)) { maxerr = e; worst = q; } if (fabs(r[q]) > maxref) maxref = fabs(r[q]); sq += e * e; sqr += r[q] * r[q]; }
            printf("check small %dx%d mode=%d grid=%d: max abs err %.3e (max |ref| %.3e, rel-L2 %.3e) worst idx %zu gpu=%f ref=%f\n", sw, sw, mode, grid, maxerr, maxref, sqrt(sq / sqr), worst, gpu[worst], r[worst]);
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(ref)); CK(hipFree(wprep));
    }
    if (legacy) for (int sw : {16, 8}) {   // timing at the training shapes: [96, 512, sw, sw] -> 512
        const int n = 96, c = 512;
        const size_t na = (size_t)n * c * sw * sw, nw = (size_t)c * c * 9;
        float *x, *w, *y; u32x4* wprep;
        CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&y, na * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wprep, nw * 4));
        fill<<<(na + 255) / 256, 256>>>(x, na, 5u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 7u, 0.1f);
        const int total = (c / TM) * (c / KC) * 9 * 2 * TM;
        hipLaunchKernelGGL(conv3x3_prep_weights, dim3((total + 255) / 256), dim3(256), 0, 0, w, wprep, c, c, 0, 3);
        conv_params p{};
        p.x = x; p.wprep = wprep; p.y = y; p.n = n; p.k = c; p.m = c; p.h = sw; p.w = sw;
        p.tiles = (n / (512 / (sw * sw))) * (c / TM); p.grid = p.tiles < 256 ? p.tiles : 256;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int it = 0; it < reps + 1; it++) {
            if (it == 1) CK(hipEventRecord(e0));
            if (sw == 16) hipLaunchKernelGGL((conv3x3_small_kernel<3, 16>), dim3(p.grid), dim3(256), small_cfg<16>::LDS, 0, p);
            else hipLaunchKernelGGL((conv3x3_small_kernel<3, 8>), dim3(p.grid), dim3(256), small_cfg<8>::LDS, 0, p);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("small 512ch %dx%d n=96 terms=3: %8.3f ms  %7.1f TFLOP/s (fp32-equivalent), %d tiles\n", sw, sw, ms, 2.0 * n * sw * sw * (double)c * c * 9 / ms / 1e9, p.tiles);
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(wprep));
    }
    struct { const char* name; int n, c, r; } shapes[] = { {"64ch 256^2", 96, 64, 256}, {"128ch 128^2", 96, 128, 128}, {"256ch 64^2", 96, 256, 64}, {"512ch 32^2", 96, 512, 32} };
    if (legacy) for (auto& s : shapes) {
        const size_t na = (size_t)s.n * s.c * s.r * s.r, nw = (size_t)s.c * s.c * 9;
        float *x, *w, *y; u32x4* wprep;
        CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&y, na * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wprep, nw * 4));
        fill<<<(na + 255) / 256, 256>>>(x, na, 5u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 7u, 0.1f);
        const double flops = 2.0 * s.n * s.r * s.r * (double)s.c * s.c * 9;
        for (int terms = 1; terms <= 3; terms += 2) for (int grid : {256, 512}) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            if (terms == 1) launch<1>(x, w, y, wprep, s.n, s.c, s.c, s.r, s.r, 0, grid); else launch<3>(x, w, y, wprep, s.n, s.c, s.c, s.r, s.r, 0, grid);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) { if (terms == 1) launch<1>(x, w, y, wprep, s.n, s.c, s.c, s.r, s.r, 0, grid); else launch<3>(x, w, y, wprep, s.n, s.c, s.c, s.r, s.r, 0, grid); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("%-12s terms=%d grid=%3d  %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)  %6.1f GB/s in+out\n", s.name, terms, grid, ms, flops / ms / 1e9, 2.0 * na * 4 / ms / 1e6);
            fflush(stdout);
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(wprep));
    }
    if (legacy) for (auto& s : shapes) {   // ---- RW = 2 variant: 8-row tiles, two workgroups per CU ----
        const size_t na = (size_t)s.n * s.c * s.r * s.r, nw = (size_t)s.c * s.c * 9;
        float *x, *w, *y; u32x4* wprep; double* ref = nullptr;
        CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&y, na * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wprep, nw * 4));
        fill<<<(na + 255) / 256, 256>>>(x, na, 5u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 7u, 0.1f);
        launch<3>(x, w, y, wprep, s.n, s.c, s.c, s.r, s.r, 0, 256);   // reference result + prepared weights
        std::vector<float> a(1 << 20), b(1 << 20);
        CK(hipMemcpy(a.data(), y + na / 2, a.size() * 4, hipMemcpyDeviceToHost));
        conv_params p{};
        p.x = x; p.wprep = wprep; p.y = y; p.n = s.n; p.k = s.c; p.m = s.c; p.h = s.r; p.w = s.r;
        p.tiles = s.n * (s.r / 8) * (s.r / SEG) * (s.c / TM);
        CK(hipFuncSetAttribute((const void*)conv3x3_kernel<3, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_rw(2)));
        for (int grid : {512, 768}) {
            p.grid = grid;
            CK(hipMemset(y, 0, na * 4));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL((conv3x3_kernel<3, 0, 2>), dim3(grid), dim3(256), lds_bytes_rw(2), 0, p);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(b.data(), y + na / 2, b.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < a.size(); i++) md = fmax(md, fabs((double)a[i] - b[i]));
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL((conv3x3_kernel<3, 0, 2>), dim3(grid), dim3(256), lds_bytes_rw(2), 0, p);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("%-12s RW=2 terms=3 grid=%3d  %8.3f ms  %7.1f TFLOP/s   max |diff| vs RW=4: %.2e\n", s.name, grid, ms, 2.0 * s.n * s.r * s.r * (double)s.c * s.c * 9 / ms / 1e9, md);
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(wprep)); (void)ref;
    }
    {   // ---- producer / consumer kernel (conv3x3_ws_kernel.h): correctness vs the naive fp64 kernel, plain and with prologue / epilogue ----
        CK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<3, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
        CK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<3, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
        CK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
        for (int mode = 0; mode < 2; mode++) {
            const int n = 3, k = 48, m = 128, h = 32, wd = 64;
            const size_t nx = (size_t)n * k * h * wd, ny = (size_t)n * m * h * wd, nw = (size_t)m * k * 9;
            float *x, *w, *y, *xsc, *osc, *bias; double* ref; u32x4* wprep;
            CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&ref, ny * 8)); CK(hipMalloc(&wprep, nw * 4));
            CK(hipMalloc(&xsc, n * k * 4)); CK(hipMalloc(&osc, n * m * 4)); CK(hipMalloc(&bias, m * 4));
            fill<<<(nx + 255) / 256, 256>>>(x, nx, 11u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 23u, 0.1f);
            fill<<<1, 256>>>(xsc, n * k, 31u, 1.f); fill<<<2, 256>>>(osc, n * m, 37u, 1.f); fill<<<1, 256>>>(bias, m, 41u, 0.5f);
            naive_conv<<<(ny + 255) / 256, 256>>>(x, w, ref, n, k, m, h, wd, mode);
            std::vector<double> r(ny); std::vector<float> gpu(ny), hx(n * k), ho(n * m), hb(m), old(ny);
            CK(hipMemcpy(r.data(), ref, ny * 8, hipMemcpyDeviceToHost));
            launch<3>(x, w, y, wprep, n, k, m, h, wd, mode, 256);   // product kernel (also prepares the weights)
            CK(hipMemcpy(old.data(), y, ny * 4, hipMemcpyDeviceToHost));
            conv_ws_params pp{};
            pp.c.x = x; pp.c.wprep = wprep; pp.c.y = y; pp.c.n = n; pp.c.k = k; pp.c.m = m; pp.c.h = h; pp.c.w = wd;
            pp.c.tiles = n * (h / TROWS) * (wd / SEG) * (m / TM);
            for (int grid : {256, 5, 1}) {
                pp.c.grid = grid < pp.c.tiles ? grid : pp.c.tiles;
                CK(hipMemset(y, 0xff, ny * 4));
                hipLaunchKernelGGL((conv3x3_ws_kernel<3, 0, 0>), dim3(pp.c.grid), dim3(512), WS_LDS_BYTES, 0, pp);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(gpu.data(), y, ny * 4, hipMemcpyDeviceToHost));
                double maxerr = 0, maxref = 0; size_t diff = 0;
                for (size_t q = 0; q < ny; q++) { double e = fabs(gpu[q] - r[q]); if (!(e <= maxerr)) maxerr = e; if (fabs(r[q]) > maxref) maxref = fabs(r[q]); if (gpu[q] != old[q]) diff++; }
                printf("check ws mode=%d grid=%d: max abs err %.3e (max |ref| %.3e); %zu of %zu outputs differ bitwise from the 4-wave kernel\n", mode, pp.c.grid, maxerr, maxref, diff, ny);
            }
            // prologue + epilogue: y = lrelu((conv(x * xs) * os + b)) * gain with the naive kernel on pre-scaled x as reference
            CK(hipMemcpy(hx.data(), xsc, n * k * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho.data(), osc, n * m * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), bias, m * 4, hipMemcpyDeviceToHost));
            float* x2; CK(hipMalloc(&x2, nx * 4));
            { std::vector<float> hx0(nx); CK(hipMemcpy(hx0.data(), x, nx * 4, hipMemcpyDeviceToHost));
              for (size_t q = 0; q < nx; q++) hx0[q] *= hx[q / ((size_t)h * wd)];
              CK(hipMemcpy(x2, hx0.data(), nx * 4, hipMemcpyHostToDevice)); }
            naive_conv<<<(ny + 255) / 256, 256>>>(x2, w, ref, n, k, m, h, wd, mode);
            CK(hipMemcpy(r.data(), ref, ny * 8, hipMemcpyDeviceToHost));
            pp.xscale = xsc; pp.oscale = osc; pp.bias = bias; pp.act = 3; pp.alpha = 0.2f; pp.gain = 1.41421356f; pp.clamp = -1.f;
            for (int grid : {256, 5}) {
                pp.c.grid = grid < pp.c.tiles ? grid : pp.c.tiles;
                CK(hipMemset(y, 0xff, ny * 4));
                hipLaunchKernelGGL((conv3x3_ws_kernel<3, 1, 1>), dim3(pp.c.grid), dim3(512), WS_LDS_BYTES, 0, pp);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(gpu.data(), y, ny * 4, hipMemcpyDeviceToHost));
                double maxerr = 0, maxref = 0;
                for (size_t q = 0; q < ny; q++) {
                    const size_t pl = q / ((size_t)h * wd);   // n * m + mm
                    double v = r[q] * ho[pl] + hb[pl % m]; v = (v > 0 ? v : 0.2 * v) * 1.41421356;
                    double e = fabs(gpu[q] - v); if (!(e <= maxerr)) maxerr = e; if (fabs(v) > maxref) maxref = fabs(v);
                }
                printf("check ws PRO=1 EPI=1 mode=%d grid=%d: max abs err %.3e (max |ref| %.3e)\n", mode, pp.c.grid, maxerr, maxref);
            }
            CK(hipFree(x)); CK(hipFree(x2)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(ref)); CK(hipFree(wprep)); CK(hipFree(xsc)); CK(hipFree(osc)); CK(hipFree(bias));
        }
    }
    for (auto& s : shapes) {   // ---- producer / consumer kernel: timing at the training shapes, interleaved with the 4-wave kernel ----
        const size_t na = (size_t)s.n * s.c * s.r * s.r, nw = (size_t)s.c * s.c * 9;
        float *x, *w, *y, *xsc, *osc, *bias; u32x4* wprep;
        CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&y, na * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wprep, nw * 4));
        CK(hipMalloc(&xsc, s.n * s.c * 4)); CK(hipMalloc(&osc, s.n * s.c * 4)); CK(hipMalloc(&bias, s.c * 4));
        fill<<<(na + 255) / 256, 256>>>(x, na, 5u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 7u, 0.1f);
        fill<<<(s.n * s.c + 255) / 256, 256>>>(xsc, s.n * s.c, 31u, 1.f); fill<<<(s.n * s.c + 255) / 256, 256>>>(osc, s.n * s.c, 37u, 1.f); fill<<<(s.c + 255) / 256, 256>>>(bias, s.c, 41u, 0.5f);
        launch<3>(x, w, y, wprep, s.n, s.c, s.c, s.r, s.r, 0, 256);
        std::vector<float> a(1 << 20), b(1 << 20);
        CK(hipMemcpy(a.data(), y + na / 2, a.size() * 4, hipMemcpyDeviceToHost));
        conv_ws_params pp{};
        pp.c.x = x; pp.c.wprep = wprep; pp.c.y = y; pp.c.n = s.n; pp.c.k = s.c; pp.c.m = s.c; pp.c.h = s.r; pp.c.w = s.r;
        pp.c.tiles = s.n * (s.r / TROWS) * (s.r / SEG) * (s.c / TM); pp.c.grid = 256;
        pp.xscale = xsc; pp.oscale = osc; pp.bias = bias; pp.act = 3; pp.alpha = 0.2f; pp.gain = 1.41421356f; pp.clamp = -1.f;
        conv_params p0 = pp.c;
        CK(hipMemset(y, 0, na * 4));
        hipLaunchKernelGGL((conv3x3_ws_kernel<3, 0, 0>), dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(b.data(), y + na / 2, b.size() * 4, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < a.size(); i++) md = fmax(md, fabs((double)a[i] - b[i]));
        const double flops = 2.0 * s.n * s.r * s.r * (double)s.c * s.c * 9;
        float best[7] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
        static bool attr2 = false;
        if (!attr2) {
            CK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<3, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
            CK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<3, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
            CK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<3, 0, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
            CK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<3, 0, 0, 0, 1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
            attr2 = true;
        }
        for (int round = 0; round < 3; round++) for (int v = 0; v < 7; v++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                if (v == 0) hipLaunchKernelGGL((conv3x3_ws_kernel<3, 0, 0, 0, 1, 0, 0>), dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);   // tap-major consumer order (ORD = 0)
                else if (v == 1) hipLaunchKernelGGL((conv3x3_ws_kernel<3, 0, 0>), dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
                else if (v == 2) hipLaunchKernelGGL((conv3x3_ws_kernel<3, 1, 1>), dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
                else if (v == 3) hipLaunchKernelGGL((conv3x3_ws_kernel<1, 0, 0>), dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
                else if (v == 4) hipLaunchKernelGGL((conv3x3_ws_kernel<3, 0, 1>), dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
                else if (v == 5) hipLaunchKernelGGL((conv3x3_ws_kernel<3, 1, 0>), dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
                else hipLaunchKernelGGL((conv3x3_ws_kernel<3, 0, 1, 4>), dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            best[v] = fminf(best[v], ms);
        }
        printf("%-12s ws ORD=0 %7.3f ms %6.1f TF | ws %7.3f ms %6.1f TF | ws+PRO+EPI %7.3f ms | ws terms=1 %7.3f ms %6.1f TF   (max |diff| ws vs 4-wave %.2e)\n", s.name,
               best[0], flops / best[0] / 1e9, best[1], flops / best[1] / 1e9, best[2], best[3], flops / best[3] / 1e9, md);
        printf("%-12s    EPI only %7.3f ms | PRO only %7.3f ms | EPI without its stores %7.3f ms\n", s.name, best[4], best[5], best[6]);
        fflush(stdout);
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(wprep)); CK(hipFree(xsc)); CK(hipFree(osc)); CK(hipFree(bias));
    }
    if (ablate) for (int si : {0, 1, 2, 3}) {   // ---- producer / consumer kernel: ablations and consumer priority (timing only where ABL != 0) ----
        auto& s = shapes[si];
        const size_t na = (size_t)s.n * s.c * s.r * s.r, nw = (size_t)s.c * s.c * 9;
        float *x, *w, *y; u32x4* wprep;
        CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&y, na * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wprep, nw * 4));
        fill<<<(na + 255) / 256, 256>>>(x, na, 5u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 7u, 0.1f);
        launch<3>(x, w, y, wprep, s.n, s.c, s.c, s.r, s.r, 0, 256);
        conv_ws_params pp{};
        pp.c.x = x; pp.c.wprep = wprep; pp.c.y = y; pp.c.n = s.n; pp.c.k = s.c; pp.c.m = s.c; pp.c.h = s.r; pp.c.w = s.r;
        pp.c.tiles = s.n * (s.r / TROWS) * (s.r / SEG) * (s.c / TM); pp.c.grid = 256;
        typedef void (*kern_t)(conv_ws_params);
        struct { const char* name; kern_t k; } abl[] = {
            {"full (consumer prio 1)", conv3x3_ws_kernel<3, 0, 0, 0, 1>},
            {"producers idle (barriers only)", conv3x3_ws_kernel<3, 0, 0, 1, 1>}, {"no MFMAs (reads + barriers)", conv3x3_ws_kernel<3, 0, 0, 2, 1>},
            {"no operand reads (MFMAs + barriers)", conv3x3_ws_kernel<3, 0, 0, 3, 1>}, {"no epilogue stores", conv3x3_ws_kernel<3, 0, 0, 4, 1>},
            {"producers without global loads / DMA", conv3x3_ws_kernel<3, 0, 0, 5, 1>}, {"producers idle + no stores", conv3x3_ws_kernel<3, 0, 0, 6, 1>},
            {"producers idle + no operand reads", conv3x3_ws_kernel<3, 0, 0, 7, 1>}, {"no weight DMA (x loads kept)", conv3x3_ws_kernel<3, 0, 0, 8, 1>},
            {"no x loads (weight DMA kept)", conv3x3_ws_kernel<3, 0, 0, 9, 1>} };
        int ai = -1;
        for (auto& a : abl) {
            ai++;
            if (!a.k || (argc > 4 && atoi(argv[4]) != ai)) continue;
            CK(hipFuncSetAttribute((const void*)a.k, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
            hipLaunchKernelGGL(a.k, dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int round = 0; round < 3; round++) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; i++) hipLaunchKernelGGL(a.k, dim3(256), dim3(512), WS_LDS_BYTES, 0, pp);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms / reps);
            }
            printf("ws ablation %-12s terms=3: %-40s %8.3f ms\n", s.name, a.name, best);
            fflush(stdout);
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(wprep));
    }
    if (legacy) {   // ---- ablations on the 128ch 128^2 layer (timing only; results are wrong by construction) ----
        const int n = 96, c = 128, r = 128;
        const size_t na = (size_t)n * c * r * r, nw = (size_t)c * c * 9;
        float *x, *w, *y; u32x4* wprep;
        CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&y, na * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wprep, nw * 4));
        fill<<<(na + 255) / 256, 256>>>(x, na, 5u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 7u, 0.1f);
        launch<3>(x, w, y, wprep, n, c, c, r, r, 0, 256);
        conv_params p{};
        p.x = x; p.wprep = wprep; p.y = y; p.n = n; p.k = c; p.m = c; p.h = r; p.w = r;
        p.tiles = n * (r / TROWS) * (r / SEG) * (c / TM); p.grid = 256;
        typedef void (*kern_t)(conv_params);
        struct { const char* name; kern_t k; } abl[] = { {"full", conv3x3_kernel<3, 0>}, {"no weight LDS copy", conv3x3_kernel<3, 1>}, {"no x split + LDS fill", conv3x3_kernel<3, 2>},
                                                         {"no global loads / fill", conv3x3_kernel<3, 3>}, {"no MFMAs / LDS reads", conv3x3_kernel<3, 4>} };
        for (auto& a : abl) {
            CK(hipFuncSetAttribute((const void*)a.k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(a.k, dim3(256), dim3(256), LDS_BYTES, 0, p);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; i++) hipLaunchKernelGGL(a.k, dim3(256), dim3(256), LDS_BYTES, 0, p);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("ablation 128ch 128^2 terms=3: %-26s %8.3f ms\n", a.name, ms);
        }
    }
    return 0;
}
