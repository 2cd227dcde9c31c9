// Templated fp32 MFMA GEMM kernel shared by gemm.hip (the product entry point) and tools/gemm_lab.hip (variant timing).
//
//   C[b][M,N] = A[b][M,K] * op(B[b]) (+ bias)          A row-major (k contiguous); B row-major [K,N] (TRANS_B=0) or [N,K] (TRANS_B=1)
//
// Workgroup = 256 threads = 4 waves as 2x2, output tile 128x128, each wave 64x64 = 2x2 v_mfma_f32_32x32x2_f32 tiles (64
// accumulator VGPRs).  K is walked in BK-deep tiles staged through LDS k-major ("[k][row]", rows padded by 4 floats: the
// MFMA operand fetch -- lanes 0-31 consecutive rows at k, lanes 32-63 at k+1 -- is a conflict-free ds_read_b32).
// Global loads of tile t+1 are issued before the MFMAs of tile t (register staging).  DBUF = 1 adds a second LDS buffer so
// one barrier per K tile suffices (the store of tile t+1 cannot overtake readers of tile t-1: they are a barrier behind).
#pragma once

#include <hip/hip_runtime.h>
#include "sgv_split.h"
#include "sgv_common.h"
#include <stdint.h>

namespace sgv_gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128;
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
    int ksplit;            // >= 1: slices of K per batch entry (k = slice length)
    const float* residual; // added to the result before the store (same layout as c), or NULL
    const float* a_amax;   // TERMS = 4 (block-scaled fp16 split, sgv_split.h): device pointers to upper bounds of max |A| and max |B| (whole tensors)
    const float* b_amax;
    float* c_amax;         // max |C| as a by-product of the store (sgv_amax_sink: a skip convolution's output is the next block's input), or NULL
};

// [128 rows x BK] block of a row-major [rows, K] matrix (k contiguous): thread t -> row t/4 (+64 per pass), k-quad t%4 (+4 per k-pass).
template <int BK> struct frag_rk { float v[2 * (BK / 16)][4]; };

template <int BK, int FULL>
__device__ __forceinline__ void load_rowmajor_k(const float* base, int64_t ld, int row0, int rows, int k0, int kdim, frag_rk<BK>& f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int kp = 0; kp < BK / 16; kp++)
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            int r = row0 + (t >> 2) + pass * 64;
            if (FULL == 2) r = min(r, rows - 1);   // rows beyond the matrix repeat its last row; their products land in accumulator rows that are never stored
            const int kq = k0 + kp * 16 + (t & 3) * 4;
            const float* p = base + (int64_t)r * ld + kq;
            const bool row_ok = r < rows;
            float* o = f.v[kp * 2 + pass];
            if (FULL || (row_ok && kq + 3 < kdim && ((((uintptr_t)p) & 15) == 0))) {
                float4 q = *(const float4*)p;
                o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = (row_ok && kq + i < kdim) ? p[i] : 0.f;
            }
        }
}

template <int BK>
__device__ __forceinline__ void store_rowmajor_k(float (*lds)[LDT], const frag_rk<BK>& f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int kp = 0; kp < BK / 16; kp++)
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int r = (t >> 2) + pass * 64;
            const int kq = kp * 16 + (t & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; i++) lds[kq + i][r] = f.v[kp * 2 + pass][i];
        }
}

// [BK x 128 cols] block of a row-major [K, cols] matrix (col contiguous): thread t -> k = t/32 (+8 per pass), col-quad t%32.
template <int BK> struct frag_kc { float v[BK / 8][4]; };

template <int BK, int FULL>
__device__ __forceinline__ void load_rowmajor_c(const float* base, int64_t ld, int col0, int cols, int k0, int kdim, frag_kc<BK>& f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < BK / 8; pass++) {
        const int kk = k0 + (t >> 5) + pass * 8;
        const int cq = col0 + (t & 31) * 4;
        const float* p = base + (int64_t)kk * ld + cq;
        const bool k_ok = kk < kdim;
        if (FULL || (k_ok && cq + 3 < cols && ((((uintptr_t)p) & 15) == 0))) {
            float4 q = *(const float4*)p;
            f.v[pass][0] = q.x; f.v[pass][1] = q.y; f.v[pass][2] = q.z; f.v[pass][3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) f.v[pass][i] = (k_ok && cq + i < cols) ? p[i] : 0.f;
        }
    }
}

template <int BK>
__device__ __forceinline__ void store_rowmajor_c(float (*lds)[LDT], const frag_kc<BK>& f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < BK / 8; pass++) {
        const int kk = (t >> 5) + pass * 8;
        const int cq = (t & 31) * 4;
        *(float4*)&lds[kk][cq] = make_float4(f.v[pass][0], f.v[pass][1], f.v[pass][2], f.v[pass][3]);
    }
}

// FULL = 1: the host guarantees m % 128 == n % 128 == k % BK == 0 and 16-B aligned rows -> no bounds or alignment branches.
// FULL = 2: the same for n and k only; m is arbitrary (the 64-row products of the discriminator's first skip branch): A rows are clamped, C rows masked.
template <int TRANS_B, int BK, int DBUF, int MINWG, int FULL>
__global__ __launch_bounds__(256, MINWG) void gemm_f32_kernel(gemm_params p) {
    __shared__ __attribute__((aligned(16))) float As[DBUF + 1][BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[DBUF + 1][BK][LDT];

    const int tile = blockIdx.x;
    const int tm = tile % p.tiles_m;  // M fastest: consecutive workgroups share the B panel
    const int tn = tile / p.tiles_m;
    const int batch = (int)blockIdx.y / p.ksplit, slice = (int)blockIdx.y - batch * p.ksplit;   // p.k is the slice length
    const float* A = p.a + batch * p.stride_a + (int64_t)slice * p.k;
    const float* B = p.b + batch * p.stride_b + (TRANS_B ? (int64_t)slice * p.k : (int64_t)slice * p.k * p.ldb);
    float* C = p.c + (int64_t)blockIdx.y * p.stride_c;
    const float* RES = p.residual ? p.residual + (int64_t)blockIdx.y * p.stride_c : nullptr;
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

    frag_rk<BK> fa;
    frag_rk<BK> fb_t;
    frag_kc<BK> fb_n;
    constexpr int FB = FULL ? 1 : 0;
    load_rowmajor_k<BK, FULL>(A, p.lda, m0, p.m, 0, p.k, fa);
    if (TRANS_B) load_rowmajor_k<BK, FB>(B, p.ldb, n0, p.n, 0, p.k, fb_t);
    else load_rowmajor_c<BK, FB>(B, p.ldb, n0, p.n, 0, p.k, fb_n);

    int cur = 0;
    for (int k0 = 0; k0 < p.k; k0 += BK) {
        if (!DBUF) __syncthreads();  // previous tile fully consumed
        store_rowmajor_k<BK>(As[cur], fa);
        if (TRANS_B) store_rowmajor_k<BK>(Bs[cur], fb_t);
        else store_rowmajor_c<BK>(Bs[cur], fb_n);
        __syncthreads();
        if (k0 + BK < p.k) {  // prefetch the next tile into registers; lands during the MFMAs below
            load_rowmajor_k<BK, FULL>(A, p.lda, m0, p.m, k0 + BK, p.k, fa);
            if (TRANS_B) load_rowmajor_k<BK, FB>(B, p.ldb, n0, p.n, k0 + BK, p.k, fb_t);
            else load_rowmajor_c<BK, FB>(B, p.ldb, n0, p.n, k0 + BK, p.k, fb_n);
        }
        // Operands of step kk+2 are fetched from LDS while the four MFMAs of step kk run.
        float a0 = As[cur][lk][wm + lr], a1 = As[cur][lk][wm + 32 + lr];
        float b0 = Bs[cur][lk][wn + lr], b1 = Bs[cur][lk][wn + 32 + lr];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < BK) {
                na0 = As[cur][kk + 2 + lk][wm + lr];
                na1 = As[cur][kk + 2 + lk][wm + 32 + lr];
                nb0 = Bs[cur][kk + 2 + lk][wn + lr];
                nb1 = Bs[cur][kk + 2 + lk][wn + 32 + lr];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (DBUF) cur ^= 1;
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
    unsigned amx = 0u;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = n0 + wn + j * 32 + lr;
            if (!FULL && col >= p.n) continue;
            const float bcol = (p.bias_mode == 1) ? p.bias[col] : 0.f;
            // the residual's 16 values of this 32x32 tile are fetched as one batch before the stores (C and the residual never alias, but the
            // compiler cannot know: interleaved, every load would wait behind the previous store)
            float rv[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                rv[e] = (RES && (FULL == 1 || row < p.m)) ? RES[(int64_t)row * p.ldc + col] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (FULL != 1 && row >= p.m) continue;
                float v = acc[i][j][e] + bcol;
                if (p.bias_mode == 2) v += p.bias[row];
                if (RES) v += rv[e];
                C[(int64_t)row * p.ldc + col] = v;
                amx = sgv_amax_fold(amx, v);
            }
        }
    if (p.c_amax) sgv_amax_commit(amx, p.c_amax);
}


// ------------------------------------------------------------------------------------------------------------------------------------
// gemm_bf16x3_kernel: the same contraction on the bf16 matrix pipe with fp32 emulation -- every fp32 operand is split v = hi + lo
// (hi = bf16_rne(v), lo = bf16_rne(v - hi)) on its way into LDS and a product is a_lo*b_hi + a_hi*b_lo + a_hi*b_hi: three
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate), the arithmetic of the 3x3 convolution family (conv3x3_kernel.h; 4.4e-6 relative error, exact on
// small integers).  Ceiling 2500 / 3 = 833 TFLOP/s fp32-equivalent against 157 of the fp32-input MFMA above: the dense 1x1 / skip GEMMs of the
// discriminator (networks.py:452) stop being bound by the fp32 matrix pipe and become streams of their operands.
//
// Same 128 x 128 tile, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles; K in chunks of 32 (two k = 16 MFMA steps), two LDS stages, one
// barrier per chunk, the global loads of chunk q + 1 in flight during the MFMAs of chunk q.  LDS layout [hi|lo][k octet][row] of 16-byte words
// (8 bf16 along k): the operand fetch of an MFMA -- lanes 0-31 consecutive rows of octet 2s, lanes 32-63 of octet 2s + 1 -- is a conflict-free
// ds_read_b128.  Fill: A ([M, K], k contiguous) and B of TRANS_B = 1 ([N, K]): a thread loads 16 consecutive k of one row (4 x 16 B) and writes two
// octets; B of TRANS_B = 0 ([K, N], n contiguous -- the NCHW activation of a 1x1 convolution): a thread gathers the 8 k of two octets for ONE
// column with dword loads that are coalesced across the wave (64 consecutive n), so that the transposition costs no LDS bank conflict.
// Requirements (host-checked): n % 128 == 0, k % 32 == 0, 16-byte aligned rows of the k-contiguous operands; any m (rows clamped, stores masked).
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));
typedef float gf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_BK = 32;                          // k per chunk
constexpr int X3_PITCH = BM + 2;                   // 16-byte words per (hi/lo, octet) plane: 128 rows + 2 (plane offsets of 8 banks: conflict-free fills)
constexpr int X3_PLANES = 2 * (X3_BK / 8);         // hi/lo x 4 octets
constexpr int X3_STAGE_WORDS = 2 * X3_PLANES * X3_PITCH;   // A and B
constexpr int X3_LDS_BYTES = 2 * X3_STAGE_WORDS * 16;      // two stages: 66,560 bytes

__device__ __forceinline__ unsigned x3_pack(float a, float b) {
    gf32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, gbf16x2));
}
__device__ __forceinline__ void x3_split8(const float* v, gu32x4& hi, gu32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const unsigned h = x3_pack(v[2 * j], v[2 * j + 1]);
        const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
        hi[j] = h;
        lo[j] = x3_pack(v[2 * j] - h0, v[2 * j + 1] - h1);
    }
}

// TERMS: 3 = bf16 split, 4 = block-scaled fp16 split (fp32-grade; a_amax / b_amax of gemm_params) -- same tile, same LDS layout, same MFMA count.
template <int TRANS_B, int TERMS = 3>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(gemm_params p) {
    extern __shared__ __attribute__((aligned(16))) gu32x4 x3_lds[];
    const int ea = sgv_conv::operand_exponent<TERMS>(p.a_amax), eb = sgv_conv::operand_exponent<TERMS>(p.b_amax);
    const float aS = sgv_conv::split_scale(ea), bS = sgv_conv::split_scale(eb);
    const int eu = sgv_conv::unscale_exponent(ea, eb);
    const int tile = blockIdx.x;
    const int tm = tile % p.tiles_m;  // M fastest: consecutive workgroups share the B panel
    const int tn = tile / p.tiles_m;
    const int batch = (int)blockIdx.y / p.ksplit, slice = (int)blockIdx.y - batch * p.ksplit;   // p.k is the slice length
    const float* A = p.a + batch * p.stride_a + (int64_t)slice * p.k;
    const float* B = p.b + batch * p.stride_b + (TRANS_B ? (int64_t)slice * p.k : (int64_t)slice * p.k * p.ldb);
    float* C = p.c + (int64_t)blockIdx.y * p.stride_c;
    const float* RES = p.residual ? p.residual + (int64_t)blockIdx.y * p.stride_c : nullptr;
    const int m0 = tm * BM, n0 = tn * BN;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // fill roles: k-contiguous operands -- thread -> (row t / 2, k half t % 2: octets 2h, 2h + 1)
    const int f_row = t >> 1, f_half = t & 1;
    const float* a_src = A + (int64_t)min(m0 + f_row, p.m - 1) * p.lda + f_half * 16;   // rows beyond the matrix repeat its last row; their products are never stored
    const float* bt_src = TRANS_B ? B + (int64_t)(n0 + f_row) * p.ldb + f_half * 16 : nullptr;
    // n-contiguous B -- thread -> (column t % 128, octet pair t / 128)
    const int g_col = t & 127, g_pair = t >> 7;
    const float* bn_src = TRANS_B ? nullptr : B + (int64_t)(g_pair * 16) * p.ldb + n0 + g_col;

    float ra[16], rb[16];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *(const float4*)(a_src + k0 + 4 * q);
            ra[4 * q] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w;
        }
        if (TRANS_B) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 v = *(const float4*)(bt_src + k0 + 4 * q);
                rb[4 * q] = v.x; rb[4 * q + 1] = v.y; rb[4 * q + 2] = v.z; rb[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) rb[e] = bn_src[(int64_t)(k0 + e) * p.ldb];
        }
    };
    auto store_chunk = [&](int stage) {
        gu32x4* as = x3_lds + stage * X3_STAGE_WORDS;
        gu32x4* bs = as + X3_PLANES * X3_PITCH;
#pragma unroll
        for (int o = 0; o < 2; o++) {
            gu32x4 hi, lo;
            sgv_conv::split8t<TERMS>(ra + 8 * o, aS, hi, lo);
            as[(0 * 4 + 2 * f_half + o) * X3_PITCH + f_row] = hi;
            as[(1 * 4 + 2 * f_half + o) * X3_PITCH + f_row] = lo;
            sgv_conv::split8t<TERMS>(rb + 8 * o, bS, hi, lo);
            if (TRANS_B) {
                bs[(0 * 4 + 2 * f_half + o) * X3_PITCH + f_row] = hi;
                bs[(1 * 4 + 2 * f_half + o) * X3_PITCH + f_row] = lo;
            } else {
                bs[(0 * 4 + 2 * g_pair + o) * X3_PITCH + g_col] = hi;
                bs[(1 * 4 + 2 * g_pair + o) * X3_PITCH + g_col] = lo;
            }
        }
    };

    load_chunk(0);
    int cur = 0;
    for (int k0 = 0; k0 < p.k; k0 += X3_BK) {
        store_chunk(cur);
        __syncthreads();     // stage `cur` is complete; the other stage's readers finished before their own barrier of the previous chunk
        if (k0 + X3_BK < p.k) load_chunk(k0 + X3_BK);
        const gu32x4* as = x3_lds + cur * X3_STAGE_WORDS;
        const gu32x4* bs = as + X3_PLANES * X3_PITCH;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            gu32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                ah[i] = as[(0 * 4 + 2 * s + lk) * X3_PITCH + wm + 32 * i + lr];
                al[i] = as[(1 * 4 + 2 * s + lk) * X3_PITCH + wm + 32 * i + lr];
                bh[i] = bs[(0 * 4 + 2 * s + lk) * X3_PITCH + wn + 32 * i + lr];
                bl[i] = bs[(1 * 4 + 2 * s + lk) * X3_PITCH + wn + 32 * i + lr];
            }
            // small terms first; three passes over the four accumulators: an MFMA never waits for the one just before it
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = sgv_conv::mma16<TERMS>(al[i], bh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = sgv_conv::mma16<TERMS>(ah[i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = sgv_conv::mma16<TERMS>(ah[i], bh[j], acc[i][j]);
        }
        cur ^= 1;
    }

    // Epilogue (as gemm_f32_kernel).  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
    unsigned amx = 0u;
    const bool full_m = m0 + BM <= p.m;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = n0 + wn + j * 32 + lr;
            const float bcol = (p.bias_mode == 1) ? p.bias[col] : 0.f;
            float rv[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                rv[e] = (RES && (full_m || row < p.m)) ? RES[(int64_t)row * p.ldc + col] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (!full_m && row >= p.m) continue;
                float v = (TERMS == 4 ? __builtin_ldexpf(acc[i][j][e], eu) : acc[i][j][e]) + bcol;
                if (p.bias_mode == 2) v += p.bias[row];
                if (RES) v += rv[e];
                C[(int64_t)row * p.ldc + col] = v;
                amx = sgv_amax_fold(amx, v);
            }
        }
    if (p.c_amax) sgv_amax_commit(amx, p.c_amax);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// gemm_bf16x3_stream_kernel: the same tile, arithmetic and LDS layout as gemm_bf16x3_kernel, as PERSISTENT workgroups (two per CU) that walk a flat
// sequence of (tile, chunk) pairs.  The skip / 1x1 products of the discriminator have K = 64 ... 512, i.e. 2 ... 16 chunks per tile: launched as one
// workgroup per tile every tile pays its own memory latency (first chunk's loads -> LDS -> MFMA -> stores, nothing of the next tile in flight meanwhile),
// and the 64 -> 128 @ 128^2 product -- 1.2 GB of operands for 26 GFLOP -- ran at 2.6-2.8 TB/s (profiles/r03_gemm_bench.log).  Here the loads of the next
// tile's first chunk are issued with the last chunk of the current tile and stay in flight across its MFMAs and its 64 epilogue stores.
// Workgroup b of G takes the tiles L, L + G, L + 2G ... with L = (b % 8) * (G / 8) + b / 8: the workgroups of one XCD (b % 8) take consecutive tiles,
// so that the tiles_m workgroups that read the same B panel share an L2.
template <int TRANS_B, int TERMS = 3>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_stream_kernel(gemm_params p, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) gu32x4 x3_lds[];
    const int ea = sgv_conv::operand_exponent<TERMS>(p.a_amax), eb = sgv_conv::operand_exponent<TERMS>(p.b_amax);
    const float aS = sgv_conv::split_scale(ea), bS = sgv_conv::split_scale(eb);
    const int eu = sgv_conv::unscale_exponent(ea, eb);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;
    const int f_row = t >> 1, f_half = t & 1;
    const int g_col = t & 127, g_pair = t >> 7;
    const int G = (int)gridDim.x;
    const int logical = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;

    struct tile_ctx { const float* a_src; const float* b_src; float* C; const float* RES; int m0, n0; };
    auto decode = [&](int tile) {
        const int per_y = p.tiles_m * p.tiles_n;
        const int y = tile / per_y, r = tile - y * per_y;
        const int tn = r / p.tiles_m, tm = r - tn * p.tiles_m;   // M fastest: consecutive tiles share the B panel
        const int batch = y / p.ksplit, slice = y - batch * p.ksplit;
        const float* A = p.a + batch * p.stride_a + (int64_t)slice * p.k;
        const float* B = p.b + batch * p.stride_b + (TRANS_B ? (int64_t)slice * p.k : (int64_t)slice * p.k * p.ldb);
        tile_ctx c;
        c.m0 = tm * BM; c.n0 = tn * BN;
        c.C = p.c + (int64_t)y * p.stride_c;
        c.RES = p.residual ? p.residual + (int64_t)y * p.stride_c : nullptr;
        c.a_src = A + (int64_t)min(c.m0 + f_row, p.m - 1) * p.lda + f_half * 16;
        c.b_src = TRANS_B ? B + (int64_t)(c.n0 + f_row) * p.ldb + f_half * 16 : B + (int64_t)(g_pair * 16) * p.ldb + c.n0 + g_col;
        return c;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    float ra[16], rb[16];
    auto load_chunk = [&](const tile_ctx& c, int k0) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *(const float4*)(c.a_src + k0 + 4 * q);
            ra[4 * q] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w;
        }
        if (TRANS_B) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 v = *(const float4*)(c.b_src + k0 + 4 * q);
                rb[4 * q] = v.x; rb[4 * q + 1] = v.y; rb[4 * q + 2] = v.z; rb[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) rb[e] = c.b_src[(int64_t)(k0 + e) * p.ldb];
        }
    };
    auto store_chunk = [&](int stage) {
        gu32x4* as = x3_lds + stage * X3_STAGE_WORDS;
        gu32x4* bs = as + X3_PLANES * X3_PITCH;
#pragma unroll
        for (int o = 0; o < 2; o++) {
            gu32x4 hi, lo;
            sgv_conv::split8t<TERMS>(ra + 8 * o, aS, hi, lo);
            as[(0 * 4 + 2 * f_half + o) * X3_PITCH + f_row] = hi;
            as[(1 * 4 + 2 * f_half + o) * X3_PITCH + f_row] = lo;
            sgv_conv::split8t<TERMS>(rb + 8 * o, bS, hi, lo);
            if (TRANS_B) {
                bs[(0 * 4 + 2 * f_half + o) * X3_PITCH + f_row] = hi;
                bs[(1 * 4 + 2 * f_half + o) * X3_PITCH + f_row] = lo;
            } else {
                bs[(0 * 4 + 2 * g_pair + o) * X3_PITCH + g_col] = hi;
                bs[(1 * 4 + 2 * g_pair + o) * X3_PITCH + g_col] = lo;
            }
        }
    };

    int tile = logical;
    if (tile >= total_tiles) return;      // (whole workgroups: every wave that continues is complete)
    unsigned amx = 0u;
    tile_ctx ct = decode(tile);
    load_chunk(ct, 0);
    int cur = 0;
    for (;;) {
        const int next = tile + G;
        const bool more = next < total_tiles;
        const tile_ctx nt = more ? decode(next) : ct;
        for (int k0 = 0; k0 < p.k; k0 += X3_BK) {
            store_chunk(cur);
            __syncthreads();     // stage `cur` is complete; the other stage's readers finished before their own barrier of the previous chunk
            if (k0 + X3_BK < p.k) load_chunk(ct, k0 + X3_BK);
            else if (more) load_chunk(nt, 0);     // the next tile's first chunk: in flight across the MFMAs below and the epilogue stores
            const gu32x4* as = x3_lds + cur * X3_STAGE_WORDS;
            const gu32x4* bs = as + X3_PLANES * X3_PITCH;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                gu32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    ah[i] = as[(0 * 4 + 2 * s + lk) * X3_PITCH + wm + 32 * i + lr];
                    al[i] = as[(1 * 4 + 2 * s + lk) * X3_PITCH + wm + 32 * i + lr];
                    bh[i] = bs[(0 * 4 + 2 * s + lk) * X3_PITCH + wn + 32 * i + lr];
                    bl[i] = bs[(1 * 4 + 2 * s + lk) * X3_PITCH + wn + 32 * i + lr];
                }
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = sgv_conv::mma16<TERMS>(al[i], bh[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = sgv_conv::mma16<TERMS>(ah[i], bl[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = sgv_conv::mma16<TERMS>(ah[i], bh[j], acc[i][j]);
            }
            cur ^= 1;
        }

        // Epilogue (as gemm_bf16x3_kernel); the accumulators are cleared for the next tile as they are stored.
        const bool full_m = ct.m0 + BM <= p.m;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = ct.n0 + wn + j * 32 + lr;
                const float bcol = (p.bias_mode == 1) ? p.bias[col] : 0.f;
                float rv[16];
                if (ct.RES) {
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int row = ct.m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                        rv[e] = (full_m || row < p.m) ? ct.RES[(int64_t)row * p.ldc + col] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = ct.m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                    float v = (TERMS == 4 ? __builtin_ldexpf(acc[i][j][e], eu) : acc[i][j][e]) + bcol;
                    acc[i][j][e] = 0.f;
                    if (!full_m && row >= p.m) continue;
                    if (p.bias_mode == 2) v += p.bias[row];
                    if (ct.RES) v += rv[e];
                    ct.C[(int64_t)row * p.ldc + col] = v;
                    amx = sgv_amax_fold(amx, v);
                }
            }
        if (!more) break;
        tile = next;
        ct = nt;
    }
    if (p.c_amax) sgv_amax_commit(amx, p.c_amax);
}


// ------------------------------------------------------------------------------------------------------------------------------------
// conv1x1_wstat_kernel: the 1x1 convolution of an NCHW activation, C[b][M, N] = W[M, K] * X[b][K, N] (M = c_out, K = c_in, N = H * W), for the shapes whose
// weight matrix fits the CU's LDS -- the discriminator's 64 -> 128 and 128 -> 256 skip products (networks.py:452) and their data gradients (W^T).
//
// These products are streams: 1.4-5 flop per byte, 1.2 GB through HBM for 26 GFLOP.  In the tiled members above every 128 x 128 tile stages BOTH operands through
// LDS behind barriers -- the activation is loaded into registers, split, written to LDS in the transposed layout, read back by four waves in lock step, and a
// tile's loads, MFMAs and 64 stores per wave run one after the other: 2.3-3.2 TB/s (profiles/r04_bench_driver_cmd.json: 8.2 ms per training step).  Here the
// roles follow the data:
//   * W (at most 128 KiB as fp16 / bf16 hi + lo) is split ONCE per workgroup into LDS, in MFMA A-operand order, and stays there: the workgroups are persistent;
//   * X never touches LDS.  The MFMA's B operand wants, per lane, 8 consecutive k (channels) of one column (pixel): lane l loads pixel n0 + (l & 31) of the
//     channels 16 s + 8 (l >> 5) + j, j = 0..7 -- dword loads that are coalesced across the half-wave (128 B of one channel row) -- and splits them in
//     registers.  The loads of a 32-pixel block are issued 32 (four k steps) at a time, two such groups in flight;
//   * a wave owns a 32-pixel block x all M rows (M / 32 accumulator tiles) and walks blocks grid-stride; the waves of a workgroup take neighbouring blocks.  There
//     is no barrier after the prologue: a wave that waits for its loads leaves its SIMD to the other waves' MFMAs and stores.
// Algorithmic bytes: 4 (K + M) N per sample; every byte crosses HBM once (M <= 256 here: one slab).
// Store: C/D layout of the 32 x 32 MFMA -- 128 contiguous bytes per half-wave and register; bias per row, residual and the magnitude bound as the tiled members.
template <int TERMS, int MT, int KS>
__global__ __launch_bounds__(512, 2) void conv1x1_wstat_kernel(gemm_params p, int px_blocks, int total_blocks) {
    constexpr int M = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) gu32x4 ws_lds[];     // [hl][k step][octet][M rows] of 16-byte words (8 k each)
    const int ea = sgv_conv::operand_exponent<TERMS>(p.a_amax), eb = sgv_conv::operand_exponent<TERMS>(p.b_amax);
    const float aS = sgv_conv::split_scale(ea), bS = sgv_conv::split_scale(eb);
    const int eu = sgv_conv::unscale_exponent(ea, eb);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), nw = (int)blockDim.x >> 6;
    const int lr = lane & 31, lk = lane >> 5;

    // prologue: the whole weight matrix, split, into LDS (rows of 8 k: two 16-byte loads)
    for (int idx = t; idx < KS * 2 * M; idx += (int)blockDim.x) {
        const int row = idx % M, oct = (idx / M) & 1, ks = idx / (2 * M);
        const float4 v0 = *(const float4*)(p.a + (int64_t)row * p.lda + ks * 16 + oct * 8);
        const float4 v1 = *(const float4*)(p.a + (int64_t)row * p.lda + ks * 16 + oct * 8 + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        gu32x4 hi, lo;
        sgv_conv::split8t<TERMS>(v, aS, hi, lo);
        ws_lds[((0 * KS + ks) * 2 + oct) * M + row] = hi;
        ws_lds[((1 * KS + ks) * 2 + oct) * M + row] = lo;
    }
    __syncthreads();

    unsigned amx = 0u;
    const int G = (int)gridDim.x;
    // addresses = wave-uniform row pointer (SGPR pair) + one 32-bit lane offset: per-load 64-bit vector addresses would cost two registers per load in flight
    const int ldb = (int)p.ldb, ldc = (int)p.ldc;                 // (host-checked: 8 * ldb and 36 * ldc elements fit 31 bits of bytes)
    const int x_lane = 8 * lk * ldb + lr, c_lane = 4 * lk * ldc + lr;
    for (int blk = (int)blockIdx.x * nw + wave; blk < total_blocks; blk += G * nw) {
        const int b = blk / px_blocks, n0 = (blk - b * px_blocks) * 32;
        const float* src = p.b + (int64_t)b * p.stride_b + n0;          // wave-uniform
        // k steps in groups of four (32 loads per lane); two groups in flight: group g + 1 is issued before group g is multiplied
        constexpr int GS = KS < 4 ? KS : 4, NG = KS / GS;
        float xv[2][GS][8];
        auto load_group = [&](int g, int buf) {
#pragma unroll
            for (int s4 = 0; s4 < GS; s4++)
#pragma unroll
                for (int j = 0; j < 8; j++) xv[buf][s4][j] = (src + (int64_t)((g * GS + s4) * 16 + j) * ldb)[x_lane];
        };
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mt][e] = 0.f;
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
#pragma unroll
            for (int s4 = 0; s4 < GS; s4++) {
                const int ks = g * GS + s4;
                gu32x4 bh, bl;
                sgv_conv::split8t<TERMS>(xv[g & 1][s4], bS, bh, bl);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const gu32x4 ah = ws_lds[((0 * KS + ks) * 2 + lk) * M + mt * 32 + lr];
                    const gu32x4 al = ws_lds[((1 * KS + ks) * 2 + lk) * M + mt * 32 + lr];
                    acc[mt] = sgv_conv::mma16<TERMS>(al, bh, acc[mt]);
                    acc[mt] = sgv_conv::mma16<TERMS>(ah, bl, acc[mt]);
                    acc[mt] = sgv_conv::mma16<TERMS>(ah, bh, acc[mt]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keeps the loads of group g + 2 and the weight reads of later groups behind this group's products (registers)
        }
        float* C = p.c + (int64_t)b * p.stride_c + n0;                  // wave-uniform
        const float* RES = p.residual ? p.residual + (int64_t)b * p.stride_c + n0 : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float rv[16];
            if (RES) {
#pragma unroll
                for (int e = 0; e < 16; e++) rv[e] = (RES + (int64_t)(mt * 32 + (e & 3) + 8 * (e >> 2)) * ldc)[c_lane];
            }
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = mt * 32 + (e & 3) + 8 * (e >> 2);        // + 4 * lk: in c_lane
                float v = TERMS == 4 ? __builtin_ldexpf(acc[mt][e], eu) : acc[mt][e];
                if (p.bias_mode == 2) v += p.bias[row + 4 * lk];
                if (RES) v += rv[e];
                (C + (int64_t)row * ldc)[c_lane] = v;
                amx = sgv_amax_fold(amx, v);
            }
        }
    }
    if (p.c_amax) sgv_amax_commit(amx, p.c_amax);
}
constexpr int wstat_lds_bytes(int mt, int ks) { return 2 * ks * 2 * 32 * mt * 16; }


}  // namespace sgv_gemm
