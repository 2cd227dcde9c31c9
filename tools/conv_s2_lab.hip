// Lab for the stride-2 3x3 kernels (stylegan-v_amd/csrc/conv3x3s2_kernel.h): check against naive fp64 kernels, then time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "conv3x3s2_kernel.h"
#include "conv3x3s2_ws_kernel.h"

using namespace sgv_conv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; p[i] = ((h & 0xffffff) / 16777216.f - 0.5f) * 2.f * scale; }
}
// strided: x [n,k,2h+1,2w+1], w [m,k,3,3] -> y [n,m,h,w]
__global__ void naive_s(const float* x, const float* w, double* y, int n, int k, int m, int h, int wd) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * m * h * wd) return;
    const int X = idx % wd, Y = (idx / wd) % h, mm = (idx / ((size_t)wd * h)) % m, nn = idx / ((size_t)wd * h * m);
    const int hin = 2 * h + 1, win = 2 * wd + 1;
    double s = 0;
    for (int kk = 0; kk < k; kk++) for (int ky = 0; ky < 3; ky++) for (int kx = 0; kx < 3; kx++)
        s += (double)w[(((size_t)mm * k + kk) * 3 + ky) * 3 + kx] * x[(((size_t)nn * k + kk) * hin + 2 * Y + ky) * win + 2 * X + kx];
    y[idx] = s;
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

static int g_abl = 0;
static int g_order = 0;
static int g_ws = 0;   // 1: the producer / consumer forms (conv3x3s2_ws_kernel.h)

template <int TERMS> static void launch(int kind, const float* x, const float* w, float* y, u32x4* wprep, int n, int k, int m, int h, int wd, int grid) {
    const int total = (m / TM) * (k / KC) * 9 * 2 * TM;
    hipLaunchKernelGGL(conv3x3_prep_weights, dim3((total + 255) / 256), dim3(256), 0, 0, w, wprep, m, k, kind == 0 ? 0 : 2, TERMS);
    s2_params p{};
    p.x = x; p.wprep = wprep; p.y = y; p.n = n; p.k = k; p.m = m; p.h = h; p.w = wd;
    p.tiles = n * (h / 8) * (wd / SEG) * (m / TM);
    p.grid = grid < p.tiles ? grid : p.tiles;
    static bool attr = false;
    if (!attr) {
        CK(hipFuncSetAttribute((const void*)conv3x3_s2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        CK(hipFuncSetAttribute((const void*)conv3x3_s2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        CK(hipFuncSetAttribute((const void*)convT3x3_s2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES));
        CK(hipFuncSetAttribute((const void*)convT3x3_s2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES));
#define SGV_ATTR(A) CK(hipFuncSetAttribute((const void*)conv3x3_s2_ws_kernel<1, A>, hipFuncAttributeMaxDynamicSharedMemorySize, S2W_LDS_BYTES)); CK(hipFuncSetAttribute((const void*)conv3x3_s2_ws_kernel<3, A>, hipFuncAttributeMaxDynamicSharedMemorySize, S2W_LDS_BYTES));
#define SGV_ATTR2(A) CK(hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<1, A>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES)); CK(hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<3, A>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES));
#define SGV_ATTR3(A) CK(hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<1, A>, hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES)); CK(hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<3, A>, hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES));
        SGV_ATTR3(0) SGV_ATTR3(6) SGV_ATTR3(7) SGV_ATTR3(8) SGV_ATTR3(10)
        SGV_ATTR2(0) SGV_ATTR2(6) SGV_ATTR2(7) SGV_ATTR2(8) SGV_ATTR2(10)
        SGV_ATTR(0) SGV_ATTR(1) SGV_ATTR(2) SGV_ATTR(3) SGV_ATTR(4) SGV_ATTR(5) SGV_ATTR(6) SGV_ATTR(7)
        attr = true;
    }
    if (kind == 0 && g_ws == 2) {
        const int words = (m / P2_TM) * (k / P2_KC) * 10 * P2_TM;
        hipLaunchKernelGGL(conv3x3_prep_weights_pairs, dim3((words + 255) / 256), dim3(256), 0, 0, w, wprep, m, k, TERMS);
        p.tiles = n * (h / P2_ROWS) * (wd / SEG) * (m / P2_TM);
        p.grid = grid < p.tiles ? grid : p.tiles;
        switch (g_abl) {
            case 0: hipLaunchKernelGGL((conv3x3_s2_pairs_kernel<TERMS, 0>), dim3(p.grid), dim3(512), P2_LDS_BYTES, 0, p, s2_epilogue{}); break;
            case 6: hipLaunchKernelGGL((conv3x3_s2_pairs_kernel<TERMS, 6>), dim3(p.grid), dim3(512), P2_LDS_BYTES, 0, p, s2_epilogue{}); break;
            case 7: hipLaunchKernelGGL((conv3x3_s2_pairs_kernel<TERMS, 7>), dim3(p.grid), dim3(512), P2_LDS_BYTES, 0, p, s2_epilogue{}); break;
            case 8: hipLaunchKernelGGL((conv3x3_s2_pairs_kernel<TERMS, 8>), dim3(p.grid), dim3(512), P2_LDS_BYTES, 0, p, s2_epilogue{}); break;
            case 10: hipLaunchKernelGGL((conv3x3_s2_pairs_kernel<TERMS, 10>), dim3(p.grid), dim3(512), P2_LDS_BYTES, 0, p, s2_epilogue{}); break;
        }
        return;
    }
    if (kind == 0 && g_ws) {
        p.tiles = n * (h / S2W_ROWS) * (wd / SEG) * (m / TM);
        p.order = g_order;
        const int units = g_order ? p.tiles / (m / TM) : p.tiles;
        p.grid = grid < units ? grid : units;
        switch (g_abl) {
            case 0: hipLaunchKernelGGL((conv3x3_s2_ws_kernel<TERMS, 0>), dim3(p.grid), dim3(512), S2W_LDS_BYTES, 0, p); break;
            case 1: hipLaunchKernelGGL((conv3x3_s2_ws_kernel<TERMS, 1>), dim3(p.grid), dim3(512), S2W_LDS_BYTES, 0, p); break;
            case 2: hipLaunchKernelGGL((conv3x3_s2_ws_kernel<TERMS, 2>), dim3(p.grid), dim3(512), S2W_LDS_BYTES, 0, p); break;
            case 3: hipLaunchKernelGGL((conv3x3_s2_ws_kernel<TERMS, 3>), dim3(p.grid), dim3(512), S2W_LDS_BYTES, 0, p); break;
            case 4: hipLaunchKernelGGL((conv3x3_s2_ws_kernel<TERMS, 4>), dim3(p.grid), dim3(512), S2W_LDS_BYTES, 0, p); break;
            case 5: hipLaunchKernelGGL((conv3x3_s2_ws_kernel<TERMS, 5>), dim3(p.grid), dim3(512), S2W_LDS_BYTES, 0, p); break;
            case 6: hipLaunchKernelGGL((conv3x3_s2_ws_kernel<TERMS, 6>), dim3(p.grid), dim3(512), S2W_LDS_BYTES, 0, p); break;
            case 7: hipLaunchKernelGGL((conv3x3_s2_ws_kernel<TERMS, 7>), dim3(p.grid), dim3(512), S2W_LDS_BYTES, 0, p); break;
        }
        return;
    }
    if (kind == 1 && g_ws) {
        p.tiles = n * (h / TW_ROWS) * (wd / SEG) * (m / TM);
        p.grid = grid < p.tiles ? grid : p.tiles;
        switch (g_abl) {
            case 6: hipLaunchKernelGGL((convT3x3_s2_ws_kernel<TERMS, 6>), dim3(p.grid), dim3(448), TW_LDS_BYTES, 0, p); break;
            case 7: hipLaunchKernelGGL((convT3x3_s2_ws_kernel<TERMS, 7>), dim3(p.grid), dim3(448), TW_LDS_BYTES, 0, p); break;
            case 8: hipLaunchKernelGGL((convT3x3_s2_ws_kernel<TERMS, 8>), dim3(p.grid), dim3(448), TW_LDS_BYTES, 0, p); break;
            case 10: hipLaunchKernelGGL((convT3x3_s2_ws_kernel<TERMS, 10>), dim3(p.grid), dim3(448), TW_LDS_BYTES, 0, p); break;
            default: hipLaunchKernelGGL((convT3x3_s2_ws_kernel<TERMS, 0>), dim3(p.grid), dim3(448), TW_LDS_BYTES, 0, p); break;
        }
    }
    if (kind == 0) hipLaunchKernelGGL(conv3x3_s2_kernel<TERMS>, dim3(p.grid), dim3(256), S_LDS_BYTES, 0, p);
    else {
        if (!g_ws) hipLaunchKernelGGL(convT3x3_s2_kernel<TERMS>, dim3(p.grid), dim3(512), T_LDS_BYTES, 0, p);
        if (getenv("NO_EDGE")) return;
        if (g_ws) {
            static float* e2 = nullptr; static size_t e2_cap = 0;
            const size_t prep = convT3x3_s2_edge_we_floats(k, m) + (size_t)n * k * h;
            if (prep > e2_cap) { if (e2) CK(hipFree(e2)); CK(hipMalloc(&e2, prep * 4)); e2_cap = prep; }
            hipLaunchKernelGGL(convT3x3_s2_edge_prep<0>, dim3((prep + 255) / 256), dim3(256), 0, 0, x, w, e2, n, k, m, h, wd);
            hipLaunchKernelGGL(convT3x3_s2_edge_mfma<0>, dim3(((h > wd ? h : wd) + 1 + 31) / 32, n * (m / 32), 2), dim3(64), 0, 0, x, e2, y, n, k, m, h, wd);
            return;
        }
        static float* edge = nullptr; static size_t edge_cap = 0;
        const size_t need = convT3x3_s2_edge_floats(n, k, m, h, wd);
        if (need > edge_cap) { if (edge) CK(hipFree(edge)); CK(hipMalloc(&edge, need * 4)); edge_cap = need; }
        hipLaunchKernelGGL(convT3x3_s2_edge_gather, dim3((need + 255) / 256), dim3(256), 0, 0, x, w, edge, n, k, m, h, wd);
        const int lmax = 2 * wd + 1 > 2 * h ? 2 * wd + 1 : 2 * h;
        hipLaunchKernelGGL(convT3x3_s2_edge_kernel, dim3((lmax + 127) / 128, n * (m / EDGE_MC), 2), dim3(128), 0, 0, edge, y, n, k, m, h, wd);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    g_ws = argc > 2 ? atoi(argv[2]) : 0;
    g_abl = argc > 3 ? atoi(argv[3]) : 0;
    const int only_kind = argc > 4 ? atoi(argv[4]) : -1;
    g_order = argc > 5 ? atoi(argv[5]) : 0;
    printf("order %d\n", g_order);
    if (g_abl) printf("ABLATION %d: results are wrong by construction\n", g_abl);
    printf("variant: %s\n", g_ws == 2 ? "producer / consumer, tap pairs, 8 rows x 128 m" : g_ws ? "producer / consumer (ws)" : "one role per wave");
    for (int kind = 0; kind < 2; kind++) {
        const int n = 2, k = 32, m = 128, h = 16, wd = 64;
        const int hb = 2 * h + 1, wb = 2 * wd + 1;
        const size_t nx = kind == 0 ? (size_t)n * k * hb * wb : (size_t)n * k * h * wd, ny = kind == 0 ? (size_t)n * m * h * wd : (size_t)n * m * hb * wb, nw = (size_t)m * k * 9;
        float *x, *w, *y; double* ref; u32x4* wprep;
        CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&ref, ny * 8)); CK(hipMalloc(&wprep, nw * 4 / 9 * 10 + 1024));
        fill<<<(nx + 255) / 256, 256>>>(x, nx, 11u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 23u, 0.1f);
        if (kind == 0) naive_s<<<(ny + 255) / 256, 256>>>(x, w, ref, n, k, m, h, wd); else naive_t<<<(ny + 255) / 256, 256>>>(x, w, ref, n, k, m, h, wd);
        std::vector<double> r(ny); std::vector<float> gpu(ny);
        CK(hipMemcpy(r.data(), ref, ny * 8, hipMemcpyDeviceToHost));
        for (int terms = 1; terms <= 3; terms += 2) for (int grid : {256, 3}) {
            CK(hipMemset(y, 0xff, ny * 4));
            if (terms == 1) launch<1>(kind, x, w, y, wprep, n, k, m, h, wd, grid); else launch<3>(kind, x, w, y, wprep, n, k, m, h, wd, grid);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(gpu.data(), y, ny * 4, hipMemcpyDeviceToHost));
            double maxerr = 0, maxref = 0, sq = 0, sqr = 0; size_t worst = 0;
            for (size_t q = 0; q < ny; q++) { double e = fabs(gpu[q] - r[q]); if (!(e <= maxerr)) { maxerr = e; worst = q; } if (fabs(r[q]) > maxref) maxref = fabs(r[q]); sq += e * e; sqr += r[q] * r[q]; }
            printf("check %s terms=%d grid=%d: max abs err %.3e (max |ref| %.3e, rel-L2 %.3e) worst idx %zu gpu=%f ref=%f\n", kind ? "transposed" : "strided", terms, grid, maxerr, maxref, sqrt(sq / sqr), worst, gpu[worst], r[worst]);
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(ref)); CK(hipFree(wprep));
    }
    // big = (2r+1)^2 tensor channels cb, small = r^2 tensor channels cs
    struct { const char* name; int n, cb, cs, r; } shapes[] = { {"256->128: 64ch|128ch", 96, 64, 128, 128}, {"128->64: 128ch|256ch", 96, 128, 256, 64}, {"64->32: 256ch|512ch", 96, 256, 512, 32}, {"32->16: 512|512 (h=16: skip)", 96, 512, 512, 16} };
    for (auto& s : shapes) {
        if (s.r < 32) continue;
        const int hb = 2 * s.r + 1;
        const size_t nbig = (size_t)s.n * s.cb * hb * hb, nsmall = (size_t)s.n * s.cs * s.r * s.r, nw = (size_t)s.cb * s.cs * 9;
        float *big, *small, *w; u32x4* wprep;
        CK(hipMalloc(&big, nbig * 4)); CK(hipMalloc(&small, nsmall * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wprep, nw * 4 / 9 * 10 + 1024));
        fill<<<(nbig + 255) / 256, 256>>>(big, nbig, 5u, 1.f); fill<<<(nsmall + 255) / 256, 256>>>(small, nsmall, 6u, 1.f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 7u, 0.1f);
        const double flops = 2.0 * s.n * s.r * s.r * (double)s.cb * s.cs * 9;
        for (int kind = 0; kind < 2; kind++) for (int terms = 1; terms <= 3; terms += 2) {
            if (only_kind >= 0 && kind != only_kind) continue;
            // D direction: strided big(cb) -> small(cs); its data gradient: transposed small(cs) -> big(cb)
            const float* x = kind == 0 ? big : small; float* y = kind == 0 ? small : big;
            const int k = kind == 0 ? s.cb : s.cs, m = kind == 0 ? s.cs : s.cb;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            if (terms == 1) launch<1>(kind, x, w, y, wprep, s.n, k, m, s.r, s.r, 256); else launch<3>(kind, x, w, y, wprep, s.n, k, m, s.r, s.r, 256);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) { if (terms == 1) launch<1>(kind, x, w, y, wprep, s.n, k, m, s.r, s.r, 256); else launch<3>(kind, x, w, y, wprep, s.n, k, m, s.r, s.r, 256); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("%-28s %-10s terms=%d  %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)  %6.1f GB/s in+out\n", s.name, kind ? "transposed" : "strided", terms, ms, flops / ms / 1e9, (nbig + nsmall) * 4.0 / ms / 1e6);
            fflush(stdout);
        }
        CK(hipFree(big)); CK(hipFree(small)); CK(hipFree(w)); CK(hipFree(wprep));
    }
    return 0;
}
