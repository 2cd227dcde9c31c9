// Dense batched fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32
// accumulate, exact fp32 -- bitwise an fmaf chain over k; 157.3 TFLOP/s chip peak).
//
//   C[b][M,N] = A[b][M,K] * op(B[b]) (+ bias)
//
// Shapes it serves (see include/sgv_ops.h): FullyConnectedLayer (trans_b=1) and 1x1 convolutions
// on NCHW (trans_b=0, A = weight shared over the batch).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA
// 32x32 tiles, 64 accumulator VGPRs), BK = 16.  A and B tiles are staged through LDS with 16-B
// global loads; LDS layout is k-major ("[k][row]") with the row dimension padded by 4 floats so the
// MFMA operand fetch (lanes 0-31: consecutive rows at k, lanes 32-63: consecutive rows at k+1) is a
// conflict-free ds_read_b32.  Global->register loads for tile t+1 are issued before the MFMAs of
// tile t (register double buffering); one barrier pair per K-tile.
//
// Algorithmic flops per launch: 2*M*N*K*batch.

#include "sgv_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = BM + 4;  // padded row length of the k-major LDS tiles

struct gemm_params {
    const float* a;
    const float* b;
    const float* bias;
    float* c;
    int m, n, k;
    int64_t lda, ldb, ldc;
    int trans_b;
    int64_t stride_a, stride_b, stride_c;
    int bias_mode;
    int tiles_m, tiles_n;
};

// Loads a [ROWS x BK] sub-block of a row-major [rows, K] matrix (k contiguous) -> per-thread registers.
// 256 threads, ROWS=128, BK=16: 2048 floats = 8 per thread = 2 x float4; thread t handles
// row = t / 4 (+64 on the second pass), k-quad = t % 4.
struct frag_rk { float v[2][4]; };

__device__ __forceinline__ void load_rowmajor_k(const float* base, int64_t ld, int row0, int rows, int k0, int kdim, frag_rk& f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int r = row0 + (t >> 2) + pass * 64;
        const int kq = k0 + (t & 3) * 4;
        const float* p = base + (int64_t)r * ld + kq;
        const bool row_ok = r < rows;
        if (row_ok && kq + 3 < kdim && ((((uintptr_t)p) & 15) == 0)) {
            float4 q = *(const float4*)p;
            f.v[pass][0] = q.x; f.v[pass][1] = q.y; f.v[pass][2] = q.z; f.v[pass][3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) f.v[pass][i] = (row_ok && kq + i < kdim) ? p[i] : 0.f;
        }
    }
}

__device__ __forceinline__ void store_rowmajor_k(float (*lds)[LDT], const frag_rk& f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int r = (t >> 2) + pass * 64;
        const int kq = (t & 3) * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) lds[kq + i][r] = f.v[pass][i];
    }
}

// Loads a [BK x COLS] sub-block of a row-major [K, cols] matrix (col contiguous): thread t handles
// k = t / 32 (+8 on the second pass), col-quad = t % 32.
struct frag_kc { float v[2][4]; };

__device__ __forceinline__ void load_rowmajor_c(const float* base, int64_t ld, int col0, int cols, int k0, int kdim, frag_kc& f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int kk = k0 + (t >> 5) + pass * 8;
        const int cq = col0 + (t & 31) * 4;
        const float* p = base + (int64_t)kk * ld + cq;
        const bool k_ok = kk < kdim;
        if (k_ok && cq + 3 < cols && ((((uintptr_t)p) & 15) == 0)) {
            float4 q = *(const float4*)p;
            f.v[pass][0] = q.x; f.v[pass][1] = q.y; f.v[pass][2] = q.z; f.v[pass][3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) f.v[pass][i] = (k_ok && cq + i < cols) ? p[i] : 0.f;
        }
    }
}

__device__ __forceinline__ void store_rowmajor_c(float (*lds)[LDT], const frag_kc& f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int kk = (t >> 5) + pass * 8;
        const int cq = (t & 31) * 4;
        *(float4*)&lds[kk][cq] = make_float4(f.v[pass][0], f.v[pass][1], f.v[pass][2], f.v[pass][3]);
    }
}

template <int TRANS_B>
__global__ __launch_bounds__(256) void gemm_f32_kernel(gemm_params p) {
    __shared__ __attribute__((aligned(16))) float As[BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDT];

    const int tile = blockIdx.x;
    const int tm = tile % p.tiles_m;  // M fastest: consecutive workgroups share the B panel
    const int tn = tile / p.tiles_m;
    const int batch = blockIdx.y;
    const float* A = p.a + batch * p.stride_a;
    const float* B = p.b + batch * p.stride_b;
    float* C = p.c + batch * p.stride_c;
    const int m0 = tm * BM, n0 = tn * BN;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    frag_rk fa;
    frag_rk fb_t;
    frag_kc fb_n;
    load_rowmajor_k(A, p.lda, m0, p.m, 0, p.k, fa);
    if (TRANS_B) load_rowmajor_k(B, p.ldb, n0, p.n, 0, p.k, fb_t);
    else load_rowmajor_c(B, p.ldb, n0, p.n, 0, p.k, fb_n);

    for (int k0 = 0; k0 < p.k; k0 += BK) {
        __syncthreads();  // previous tile fully consumed
        store_rowmajor_k(As, fa);
        if (TRANS_B) store_rowmajor_k(Bs, fb_t);
        else store_rowmajor_c(Bs, fb_n);
        __syncthreads();
        if (k0 + BK < p.k) {  // prefetch the next tile into registers; lands during the MFMAs below
            load_rowmajor_k(A, p.lda, m0, p.m, k0 + BK, p.k, fa);
            if (TRANS_B) load_rowmajor_k(B, p.ldb, n0, p.n, k0 + BK, p.k, fb_t);
            else load_rowmajor_c(B, p.ldb, n0, p.n, k0 + BK, p.k, fb_n);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0 = As[kk + lk][wm + lr];
            float a1 = As[kk + lk][wm + 32 + lr];
            float b0 = Bs[kk + lk][wn + lr];
            float b1 = Bs[kk + lk][wn + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = n0 + wn + j * 32 + lr;
            if (col >= p.n) continue;
            const float bcol = (p.bias_mode == 1) ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (row >= p.m) continue;
                float v = acc[i][j][e] + bcol;
                if (p.bias_mode == 2) v += p.bias[row];
                C[(int64_t)row * p.ldc + col] = v;
            }
        }
}

}  // namespace

extern "C" int sgv_gemm_f32(const sgv_gemm_params* p, void* stream_) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: params is NULL");
    if (!p->a || !p->b || !p->c) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: NULL pointer");
    if (p->m < 1 || p->n < 1 || p->k < 1 || p->batch < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: sizes must be positive");
    if (p->bias_mode < 0 || p->bias_mode > 2 || (p->bias_mode && !p->bias)) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: bad bias_mode");
    if (p->batch > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "gemm: batch is too large");
    hipStream_t stream = (hipStream_t)stream_;
    gemm_params gp;
    gp.a = p->a; gp.b = p->b; gp.bias = p->bias; gp.c = p->c;
    gp.m = p->m; gp.n = p->n; gp.k = p->k;
    gp.lda = p->lda; gp.ldb = p->ldb; gp.ldc = p->ldc;
    gp.trans_b = p->trans_b;
    gp.stride_a = p->stride_a; gp.stride_b = p->stride_b; gp.stride_c = p->stride_c;
    gp.bias_mode = p->bias_mode;
    gp.tiles_m = (p->m + BM - 1) / BM;
    gp.tiles_n = (p->n + BN - 1) / BN;
    const double flops = 2.0 * p->m * p->n * p->k * p->batch;
    const double bytes = 4.0 * ((double)p->m * p->k * (p->stride_a ? p->batch : 1) + (double)p->n * p->k * (p->stride_b ? p->batch : 1) +
                                (double)p->m * p->n * p->batch);
    sgv_launch_scope scope(SGV_K_GEMM, stream, bytes, flops);
    dim3 grid((unsigned)(gp.tiles_m * gp.tiles_n), (unsigned)p->batch);
    if (p->trans_b) hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, dim3(256), 0, stream, gp);
    else hipLaunchKernelGGL(gemm_f32_kernel<0>, grid, dim3(256), 0, stream, gp);
    return sgv_check_launch("gemm_f32_kernel");
}
