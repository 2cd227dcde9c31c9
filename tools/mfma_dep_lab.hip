// How far apart must two v_mfma_f32_32x32x16_bf16 on the SAME accumulator be issued to run at the pipe's rate?
// One wave per SIMD (256 workgroups x 256 threads), D accumulators used round-robin, 4096 MFMAs per wave.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_dep_lab.hip -o tools/mfma_dep_lab && tools/mfma_dep_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int D>
__global__ __launch_bounds__(256) void chain(float* out, int iters) {
    f32x16 acc[D];
    for (int d = 0; d < D; d++) for (int e = 0; e < 16; e++) acc[d][e] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.f + i * 0.01f); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int d = 0; d < D; d++) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[d], 0, 0, 0);
    }
    float s = 0.f;
    for (int d = 0; d < D; d++) for (int e = 0; e < 16; e++) s += acc[d][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int D> int run(float* out) {
    const int per_iter = 8 * D, iters = 4096 / per_iter;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chain<D>, dim3(256), dim3(256), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(chain<D>, dim3(256), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    const double mfmas = (double)iters * per_iter;
    printf("reuse distance %d: %.1f ns per MFMA per wave (%.0f TFLOP/s bf16 over the chip)\n", D, ms * 1e6 / mfmas, 1024.0 * mfmas * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 256 * 4));
    run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<6>(out); run<8>(out);
    return 0;
}
