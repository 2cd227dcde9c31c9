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
#include "gemm_kernel.h"
#include <stdlib.h>
#include <mutex>
#include <set>

using namespace sgv_gemm;

static inline int slice_ok(const sgv_gemm_params*, int) { return 1; }   // k % 32 == 0 keeps every K slice 16-byte aligned

extern "C" int sgv_gemm_f32(const sgv_gemm_params* p, void* stream_) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: params is NULL");
    if (!p->a || !p->b || !p->c) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: NULL pointer");
    if (p->m < 1 || p->n < 1 || p->k < 1 || p->batch < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: sizes must be positive");
    if (p->bias_mode < 0 || p->bias_mode > 2 || (p->bias_mode && !p->bias)) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: bad bias_mode");
    const int ks = p->k_split > 1 ? p->k_split : 1;
    if (ks > 1 && (p->k % ks != 0 || p->bias_mode != 0 || p->residual)) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: k_split must divide k and excludes a bias / residual");
    if ((int64_t)p->batch * ks > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "gemm: batch (x k_split) is too large");
    hipStream_t stream = (hipStream_t)stream_;
    gemm_params gp;
    gp.a = p->a; gp.b = p->b; gp.bias = p->bias; gp.c = p->c;
    gp.m = p->m; gp.n = p->n; gp.k = p->k / ks; gp.ksplit = ks;
    gp.lda = p->lda; gp.ldb = p->ldb; gp.ldc = p->ldc;
    gp.trans_b = p->trans_b;
    gp.stride_a = p->stride_a; gp.stride_b = p->stride_b; gp.stride_c = p->stride_c;
    gp.bias_mode = p->bias_mode;
    gp.residual = p->residual;
    // exact_fp32 = 2: the split members with block-scaled fp16 operands (fp32-grade); falls back to the exact fp32 pipe where the shape rules them out
    const bool f16 = p->exact_fp32 == 2;
    if (f16 && (!p->a_amax || !p->b_amax)) return sgv_fail(SGV_ERR_INVALID_ARG, "gemm: exact_fp32 = 2 (block-scaled fp16 split) needs a_amax and b_amax (sgv_absmax)");
    gp.a_amax = f16 ? p->a_amax : nullptr; gp.b_amax = f16 ? p->b_amax : nullptr;
    gp.tiles_m = (p->m + BM - 1) / BM;
    gp.tiles_n = (p->n + BN - 1) / BN;
    const double flops = 2.0 * p->m * p->n * p->k * p->batch;
    const double bytes = 4.0 * ((double)p->m * p->k * (p->stride_a ? p->batch : 1) + (double)p->n * p->k * (p->stride_b ? p->batch : 1) +
                                (double)p->m * p->n * p->batch);
    sgv_launch_scope scope(SGV_K_GEMM, stream, bytes, flops);
    gp.c_amax = ks == 1 ? scope.take_amax_sink() : nullptr;      // (K slices are partial sums: no bound of the result there)
    dim3 grid((unsigned)(gp.tiles_m * gp.tiles_n), (unsigned)(p->batch * ks));
    // Fast path (tools/gemm_lab.hip, profiles/r01_gemm_lab.log): whole tiles, 16-B aligned rows -> no bounds/alignment branches in
    // the K loop; 1x1 convolutions run bk16 + double-buffered LDS at 2 workgroups/CU, x @ w.T runs bk32.
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool full16 = p->m % BM == 0 && p->n % BN == 0 && gp.k % 16 == 0 && p->lda % 4 == 0 && p->ldb % 4 == 0 && p->stride_a % 4 == 0 &&
                        p->stride_b % 4 == 0 && al16(p->a) && al16(p->b);
    static const bool allow_nk = !(getenv("SGV_GEMM_FULLNK") && getenv("SGV_GEMM_FULLNK")[0] == '0');
    const bool full_nk = allow_nk && !full16 && p->n % BN == 0 && gp.k % 16 == 0 && p->lda % 4 == 0 && p->ldb % 4 == 0 && p->stride_a % 4 == 0 && p->stride_b % 4 == 0 &&
                         al16(p->a) && al16(p->b);
    // bf16x3 member (gemm_kernel.h): whole tiles along n, k % 32 == 0, 16-byte aligned k-contiguous rows; any m.  SGV_GEMM_TERMS=0 keeps every
    // product on the exact-fp32 matrix pipe.
    static const bool allow_x3 = !(getenv("SGV_GEMM_TERMS") && getenv("SGV_GEMM_TERMS")[0] == '0');
    const bool x3 = allow_x3 && p->exact_fp32 != 1 && p->n % BN == 0 && gp.k % X3_BK == 0 && p->lda % 4 == 0 && p->stride_a % 4 == 0 && al16(p->a) && (int64_t)slice_ok(p, ks) &&
                    (!p->trans_b || (p->ldb % 4 == 0 && p->stride_b % 4 == 0 && al16(p->b)));
    // W-stationary member (conv1x1_wstat_kernel): a 1x1 convolution on NCHW (shared A, n-contiguous B) whose weight matrix fits LDS -- the 64 <-> 128 and
    // 128 <-> 256 channel skip products and their data gradients; SGV_GEMM_WSTAT=0 keeps them on the tiled members.
    static const bool allow_wstat = !(getenv("SGV_GEMM_WSTAT") && getenv("SGV_GEMM_WSTAT")[0] == '0');
    if (x3 && allow_wstat && !p->trans_b && p->stride_a == 0 && ks == 1 && p->bias_mode != 1 && p->m % 32 == 0 && p->k % 16 == 0 && p->n % 32 == 0) {
        const int mt = p->m / 32, kst = p->k / 16;
        const int64_t blocks = (int64_t)p->batch * (p->n / 32);
        typedef void (*wstat_fn)(gemm_params, int, int);
        wstat_fn fn = nullptr;
#define SGV_WSTAT(MT_, KS_) if (mt == MT_ && kst == KS_) fn = f16 ? (wstat_fn)conv1x1_wstat_kernel<4, MT_, KS_> : (wstat_fn)conv1x1_wstat_kernel<3, MT_, KS_>;
        SGV_WSTAT(4, 4) SGV_WSTAT(2, 8) SGV_WSTAT(8, 8) SGV_WSTAT(4, 16)
#undef SGV_WSTAT
        static const int cus = [] {
            int dev = 0, n = 256;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
            return n;
        }();
        if (fn && blocks >= 8 * cus && blocks <= INT32_MAX && p->ldb < (1 << 26) && p->ldc < (1 << 26)) {   // (the prologue reads the whole weight matrix per workgroup; 32-bit lane offsets)
            const int lds = wstat_lds_bytes(mt, kst);
            static std::mutex mu;
            static std::set<const void*> prepared;
            {
                std::lock_guard<std::mutex> lock(mu);
                if (!prepared.count((const void*)fn)) {
                    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "gemm: hipFuncSetAttribute failed");
                    prepared.insert((const void*)fn);
                }
            }
            // 32-KiB weight images (130-148 registers): two workgroups of six waves per CU = three waves per SIMD; 128-KiB images: one workgroup of eight
            if (lds <= 64 * 1024) hipLaunchKernelGGL(fn, dim3((unsigned)(2 * cus)), dim3(384), lds, stream, gp, p->n / 32, (int)blocks);
            else
            hipLaunchKernelGGL(fn, dim3((unsigned)cus), dim3(512), lds, stream, gp, p->n / 32, (int)blocks);
            sgv_note_variant(SGV_V_conv1x1_wstat);
            return sgv_check_launch("conv1x1_wstat_kernel");
        }
    }
    if (x3) {
        static const hipError_t attr_err = [] {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
            return e;
        }();
        if (attr_err != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
        // Persistent form (gemm_bf16x3_stream_kernel: two workgroups per CU walk the tiles, the next tile's first chunk in flight across the epilogue)
        // wherever the launch has more tiles than resident workgroups; SGV_GEMM_STREAM=0 keeps one workgroup per tile.
        static const bool allow_stream = !(getenv("SGV_GEMM_STREAM") && getenv("SGV_GEMM_STREAM")[0] == '0');
        static const int resident = [] {
            int dev = 0, cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
            return 2 * cus;
        }();
        const int64_t total_tiles = (int64_t)gp.tiles_m * gp.tiles_n * p->batch * ks;
        if (allow_stream && total_tiles > resident && total_tiles <= INT32_MAX) {
            static const hipError_t attr_err_s = [] {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16x3_stream_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
                if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16x3_stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
                if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16x3_stream_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
                if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16x3_stream_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
                return e;
            }();
            if (attr_err_s != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err_s));
            const dim3 sgrid((unsigned)(resident & ~7));
            if (f16) {
                if (p->trans_b) hipLaunchKernelGGL((gemm_bf16x3_stream_kernel<1, 4>), sgrid, dim3(256), X3_LDS_BYTES, stream, gp, (int)total_tiles);
                else hipLaunchKernelGGL((gemm_bf16x3_stream_kernel<0, 4>), sgrid, dim3(256), X3_LDS_BYTES, stream, gp, (int)total_tiles);
            } else if (p->trans_b) hipLaunchKernelGGL(gemm_bf16x3_stream_kernel<1>, sgrid, dim3(256), X3_LDS_BYTES, stream, gp, (int)total_tiles);
            else hipLaunchKernelGGL(gemm_bf16x3_stream_kernel<0>, sgrid, dim3(256), X3_LDS_BYTES, stream, gp, (int)total_tiles);
            sgv_note_variant(SGV_V_gemm_bf16x3_stream);
            return sgv_check_launch("gemm_bf16x3_stream_kernel");
        }
        if (f16) {
            if (p->trans_b) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 4>), grid, dim3(256), X3_LDS_BYTES, stream, gp);
            else hipLaunchKernelGGL((gemm_bf16x3_kernel<0, 4>), grid, dim3(256), X3_LDS_BYTES, stream, gp);
        } else if (p->trans_b) hipLaunchKernelGGL(gemm_bf16x3_kernel<1>, grid, dim3(256), X3_LDS_BYTES, stream, gp);
        else hipLaunchKernelGGL(gemm_bf16x3_kernel<0>, grid, dim3(256), X3_LDS_BYTES, stream, gp);
        sgv_note_variant(SGV_V_gemm_bf16x3);
        return sgv_check_launch("gemm_bf16x3_kernel");
    }
    if (full_nk) {   // whole tiles along n and k, any m
        if (p->trans_b) hipLaunchKernelGGL((gemm_f32_kernel<1, 16, 1, 2, 2>), grid, dim3(256), 0, stream, gp);
        else hipLaunchKernelGGL((gemm_f32_kernel<0, 16, 1, 2, 2>), grid, dim3(256), 0, stream, gp);
    } else if (p->trans_b) {
        if (full16 && gp.k % 32 == 0) hipLaunchKernelGGL((gemm_f32_kernel<1, 32, 0, 1, 1>), grid, dim3(256), 0, stream, gp);
        else if (full16) hipLaunchKernelGGL((gemm_f32_kernel<1, 16, 1, 2, 1>), grid, dim3(256), 0, stream, gp);
        else hipLaunchKernelGGL((gemm_f32_kernel<1, 16, 0, 1, 0>), grid, dim3(256), 0, stream, gp);
    } else {
        if (full16) hipLaunchKernelGGL((gemm_f32_kernel<0, 16, 1, 2, 1>), grid, dim3(256), 0, stream, gp);
        else hipLaunchKernelGGL((gemm_f32_kernel<0, 16, 0, 1, 0>), grid, dim3(256), 0, stream, gp);
    }
    sgv_note_variant(SGV_V_gemm_f32);
    return sgv_check_launch("gemm_f32_kernel");
}
