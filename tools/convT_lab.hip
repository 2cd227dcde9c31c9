// Lab for the consumer mapping of convT3x3_s2_ws_kernel (stylegan-v_amd/csrc/conv3x3s2_ws_kernel.h, template parameter CM): check both mappings against a
// naive fp64 kernel on the interior (the last output row / column belong to the edge kernel), then time them side by side on the benchmark's layer shapes,
// whole kernel and ablations (7: consumers alone, 8: no stores, 6: producers + DMA alone).
//   hipcc --offload-arch=gfx950 -O3 -I stylegan-v_amd/csrc -I include tools/convT_lab.hip -o tools/convT_lab
//   tools/convT_lab [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "sgv_common.h"
#include "conv3x3s2_kernel.h"
#include "conv3x3s2_ws_kernel.h"

using namespace sgv_conv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; p[i] = ((h & 0xffffff) / 16777216.f - 0.5f) * 2.f * scale; }
}
// transposed: x [n,k,h,w], w [k,m,3,3] -> y [n,m,2h+1,2w+1]
__global__ void naive_t(const float* x, const float* w, double* y, int n, int k, int m, int h, int wd) {
    const int hout = 2 * h + 1, wout = 2 * wd + 1;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * m * hout * wout) return;
    const int ox = idx % wout, oy = (idx / wout) % hout, mm = (idx / ((size_t)wout * hout)) % m, nn = idx / ((size_t)wout * hout * m);
    double s = 0;
    for (int ky = 0; ky < 3; ky++) { const int ty = oy - ky; if (ty < 0 || (ty & 1) || ty / 2 >= h) continue;
        for (int kx = 0; kx < 3; kx++) { const int tx = ox - kx; if (tx < 0 || (tx & 1) || tx / 2 >= wd) continue;
            for (int kk = 0; kk < k; kk++) s += (double)w[(((size_t)kk * m + mm) * 3 + ky) * 3 + kx] * x[(((size_t)nn * k + kk) * h + ty / 2) * wd + tx / 2]; } }
    y[idx] = s;
}

static float *g_xa, *g_wa;   // bounds of max |x|, max |w| (terms = 4)

template <int TERMS, int ABL, int S, int CM, int ST = 0>
static void go(const s2_params& p) {
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<TERMS, ABL, S, 0, CM, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(S))); attr = true; }
    hipLaunchKernelGGL((convT3x3_s2_ws_kernel<TERMS, ABL, S, 0, CM, ST>), dim3(p.grid), dim3(448), tw_lds_bytes(S), 0, p);
}

template <int TERMS, int S>
static void launch(int cm, int abl, const float* x, const float* w, float* y, u32x4* wprep, int n, int k, int m, int h, int wd, int grid) {
    const int total = ((m + TM - 1) / TM) * (k / KC) * 9 * 2 * TM;
    hipLaunchKernelGGL(conv3x3_prep_weights, dim3((total + 255) / 256), dim3(256), 0, 0, w, wprep, m, k, 2, TERMS, g_wa);
    s2_params p{};
    p.x = x; p.wprep = wprep; p.y = y; p.n = n; p.k = k; p.m = m; p.h = h; p.w = wd; p.x_amax = g_xa; p.w_amax = g_wa;
    p.tiles = S == 1 ? n * (h / TW_ROWS) * (wd / SEG) * ((m + TM - 1) / TM) : (n / S) * (h / TW_ROWS) * (m / TM);
    p.grid = grid < p.tiles ? grid : p.tiles;
    if (cm == 3 && (k / KC < 2 || m % TM != 0)) cm = 0;      // the folded-store form: >= 2 chunks, whole 64-channel tiles (as the host dispatch)
#define CASE(A) case A: if (cm == 3) go<TERMS, A, S, 0, 1>(p); else if (cm == 2) go<TERMS, A, S, 2>(p); else if (cm) go<TERMS, A, S, 1>(p); else go<TERMS, A, S, 0>(p); break;
    switch (abl) { CASE(0) CASE(6) CASE(7) CASE(8) }
#undef CASE
}

template <int S>
static void check(int n, int k, int m, int h, int wd) {
    const int hb = 2 * h + 1, wb = 2 * wd + 1;
    const size_t nx = (size_t)n * k * h * wd, ny = (size_t)n * m * hb * wb, nw = (size_t)m * k * 9;
    float *x, *w, *y; double* ref; u32x4* wprep;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&ref, ny * 8)); CK(hipMalloc(&wprep, (size_t)((m + TM - 1) / TM) * TM * k * 9 * 4 + 1024));
    fill<<<(nx + 255) / 256, 256>>>(x, nx, 11u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 23u, 0.1f);
    naive_t<<<(ny + 255) / 256, 256>>>(x, w, ref, n, k, m, h, wd);
    std::vector<double> r(ny); std::vector<float> gpu(ny);
    CK(hipMemcpy(r.data(), ref, ny * 8, hipMemcpyDeviceToHost));
    for (int cm : {0, 3}) for (int terms : {1, 3, 4}) for (int grid : {256, 3}) {
        CK(hipMemset(y, 0xff, ny * 4));
        if (terms == 1) launch<1, S>(cm, 0, x, w, y, wprep, n, k, m, h, wd, grid); else if (terms == 3) launch<3, S>(cm, 0, x, w, y, wprep, n, k, m, h, wd, grid); else launch<4, S>(cm, 0, x, w, y, wprep, n, k, m, h, wd, grid);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(gpu.data(), y, ny * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0, sq = 0, sqr = 0; size_t worst = 0, bad = 0;
        for (size_t q = 0; q < ny; q++) {
            const int ox = q % wb, oy = (q / wb) % hb;
            if (ox == 2 * wd || oy == 2 * h) continue;       // the edge kernel's outputs
            double e = fabs(gpu[q] - r[q]); if (!(e <= maxerr)) { maxerr = e; worst = q; } if (fabs(r[q]) > maxref) maxref = fabs(r[q]); sq += e * e; sqr += r[q] * r[q];
            if (!(e <= 1e-2 * 2.0)) bad++;
        }
        printf("check S=%d n=%d k=%d m=%d %dx%d CM=%d terms=%d grid=%d: max abs err %.3e (max |ref| %.3e, rel-L2 %.3e) bad %zu worst idx %zu gpu=%f ref=%f\n", S, n, k, m, h, wd, cm, terms, grid, maxerr, maxref,
               sqrt(sq / sqr), bad, worst, gpu[worst], r[worst]);
    }
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(ref)); CK(hipFree(wprep));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    CK(hipMalloc(&g_xa, 4)); CK(hipMalloc(&g_wa, 4));
    const float xa = 1.f, wa = 0.1f;
    CK(hipMemcpy(g_xa, &xa, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g_wa, &wa, 4, hipMemcpyHostToDevice));
    check<1>(2, 32, 128, 16, 64);
    check<1>(1, 16, 96, 8, 32);       // a half-full last m tile
    check<2>(4, 32, 64, 16, 16);      // packed samples
    check<4>(4, 16, 64, 8, 8);
    // small = r^2 tensor with cs channels -> big (2r+1)^2 tensor with cb channels
    struct { const char* name; int n, cb, cs, r; } shapes[] = { {"128->256: 128ch -> 64ch", 96, 64, 128, 128}, {"64->128: 256ch -> 128ch", 96, 128, 256, 64}, {"32->64: 512ch -> 256ch", 96, 256, 512, 32} };
    for (auto& s : shapes) {
        const int hb = 2 * s.r + 1;
        const size_t nbig = (size_t)s.n * s.cb * hb * hb, nsmall = (size_t)s.n * s.cs * s.r * s.r, nw = (size_t)s.cb * s.cs * 9;
        float *big, *small, *w; u32x4* wprep;
        CK(hipMalloc(&big, nbig * 4)); CK(hipMalloc(&small, nsmall * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wprep, nw * 4 + 1024));
        fill<<<(nbig + 255) / 256, 256>>>(big, nbig, 5u, 1.f); fill<<<(nsmall + 255) / 256, 256>>>(small, nsmall, 6u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 7u, 0.1f);
        const double flops = 2.0 * s.n * s.r * s.r * (double)s.cb * s.cs * 9;
        for (int abl : {0, 7, 8}) for (int terms : {3, 4}) for (int cm : {0, 3}) {     // producers + DMA alone: the mapping does not enter
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto one = [&]() { if (terms == 3) launch<3, 1>(cm, abl, small, w, big, wprep, s.n, s.cs, s.cb, s.r, s.r, 256); else launch<4, 1>(cm, abl, small, w, big, wprep, s.n, s.cs, s.cb, s.r, s.r, 256); };
            one();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) one();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("%-26s ABL=%-2d terms=%d CM=%d (3 = stores folded into the MFMA stream)  %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)  %6.1f GB/s in+out\n", s.name, abl, terms, cm, ms, flops / ms / 1e9, (nbig + nsmall) * 4.0 / ms / 1e6);
            fflush(stdout);
        }
        CK(hipFree(big)); CK(hipFree(small)); CK(hipFree(w)); CK(hipFree(wprep));
    }
    return 0;
}
