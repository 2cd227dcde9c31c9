// Variant timing for the fp32 MFMA GEMM (stylegan-v_amd/csrc/gemm_kernel.h) on the hot 1x1-convolution shapes.
// Standalone: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Istylegan-v_amd/csrc tools/gemm_lab.hip -o tools/gemm_lab && tools/gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "gemm_kernel.h"

using namespace sgv_gemm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct shape { const char* name; int batch, m, n, k, trans_b, a_shared; };

typedef void (*kern_t)(gemm_params);
struct variant { const char* name; kern_t k0, k1; int bk; int full; };

#define VAR(NAME, BK, DBUF, MINWG, FULL) { NAME, gemm_f32_kernel<0, BK, DBUF, MINWG, FULL>, gemm_f32_kernel<1, BK, DBUF, MINWG, FULL>, BK, FULL }

static variant variants[] = {
    VAR("generic bk16", 16, 0, 1, 0),
    VAR("full bk16", 16, 0, 1, 1),
    VAR("full bk16 dbuf", 16, 1, 1, 1),
    VAR("full bk32", 32, 0, 1, 1),
    VAR("full bk32 dbuf", 32, 1, 1, 1),
    VAR("full bk16 wg2", 16, 0, 2, 1),
    VAR("full bk16 dbuf wg2", 16, 1, 2, 1),
    VAR("full bk32 dbuf wg2", 32, 1, 2, 1),
    VAR("full bk16 dbuf wg3", 16, 1, 3, 1),
};

__global__ void fill(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = ((h & 0xffff) / 65536.f - 0.5f); }
}

__global__ void maxdiff(const float* a, const float* b, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float d = fabsf(a[i] - b[i]); if (d > 0) atomicMax((int*)out, __float_as_int(d)); }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    shape shapes[] = {
        {"D b64 skip 1x1  [96,256,32,32]x[512,256]", 96, 512, 1024, 256, 0, 1},
        {"D b128 skip 1x1 [96,128,64,64]x[256,128]", 96, 256, 4096, 128, 0, 1},
        {"D b32 skip 1x1  [96,512,16,16]x[512,512]", 96, 512, 256, 512, 0, 1},
        {"square 4096^3 (x @ w.T)", 1, 4096, 4096, 4096, 1, 0},
    };
    const int NSET = 3;
    for (const shape& s : shapes) {
        const size_t na = (size_t)s.m * s.k * (s.a_shared ? 1 : s.batch), nb = (size_t)s.n * s.k * s.batch, nc = (size_t)s.m * s.n * s.batch;
        float *a, *b[NSET], *c[NSET], *cref, *dmax;
        CK(hipMalloc(&a, na * 4)); CK(hipMalloc(&cref, nc * 4)); CK(hipMalloc(&dmax, 4));
        for (int i = 0; i < NSET; i++) { CK(hipMalloc(&b[i], nb * 4)); CK(hipMalloc(&c[i], nc * 4)); fill<<<(nb + 255) / 256, 256>>>(b[i], nb, 77u); }
        fill<<<(na + 255) / 256, 256>>>(a, na, 1234u);
        printf("== %s   %.1f GFLOP, %.0f MB in + %.0f MB out\n", s.name, 2.0 * s.m * s.n * s.k * s.batch / 1e9, (na + nb) * 4 / 1e6, nc * 4 / 1e6);
        for (size_t vi = 0; vi < sizeof(variants) / sizeof(variants[0]); vi++) {
            const variant& v = variants[vi];
            gemm_params p{};
            p.a = a; p.bias = nullptr; p.m = s.m; p.n = s.n; p.k = s.k;
            p.lda = s.k; p.ldb = s.trans_b ? s.k : s.n; p.ldc = s.n; p.trans_b = s.trans_b;
            p.stride_a = s.a_shared ? 0 : (int64_t)s.m * s.k; p.stride_b = (int64_t)s.n * s.k; p.stride_c = (int64_t)s.m * s.n;
            p.bias_mode = 0; p.tiles_m = (s.m + BM - 1) / BM; p.tiles_n = (s.n + BN - 1) / BN;
            if (v.full && (s.m % BM || s.n % BN || s.k % v.bk)) continue;
            dim3 grid(p.tiles_m * p.tiles_n, s.batch);
            kern_t k = s.trans_b ? v.k1 : v.k0;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int r = 0; r < 3; r++) { p.b = b[r % NSET]; p.c = c[r % NSET]; hipLaunchKernelGGL(k, grid, dim3(256), 0, 0, p); }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) { p.b = b[r % NSET]; p.c = c[r % NSET]; hipLaunchKernelGGL(k, grid, dim3(256), 0, 0, p); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            float d = 0.f;
            if (vi == 0) CK(hipMemcpy(cref, c[0], nc * 4, hipMemcpyDeviceToDevice));
            else { CK(hipMemset(dmax, 0, 4)); maxdiff<<<(nc + 255) / 256, 256>>>(c[0], cref, nc, dmax); CK(hipMemcpy(&d, dmax, 4, hipMemcpyDeviceToHost)); }
            const double tf = 2.0 * s.m * s.n * s.k * s.batch / (ms * 1e-3) / 1e12;
            printf("   %-22s %8.1f us  %6.1f TFLOP/s  %4.1f%% of 157.3   maxdiff vs generic %.2e\n", v.name, ms * 1e3, tf, 100 * tf / 157.3, d);
            fflush(stdout);
        }
        CK(hipFree(a)); CK(hipFree(cref)); CK(hipFree(dmax));
        for (int i = 0; i < NSET; i++) { CK(hipFree(b[i])); CK(hipFree(c[i])); }
    }
    return 0;
}
