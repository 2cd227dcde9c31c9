// The stride-2 members of the 3x3 convolution family on NCHW fp32 tensors (bf16x3 fp32 emulation, see wrw_kernel.h):
//
//   conv3x3_s2_kernel   y[n,m,Y,X]       = sum_{k,ky,kx} wgt(m,k,ky,kx) * x[n,k,2Y+ky,2X+kx]       x: (2H+1)x(2W+1) -> y: HxW
//   convT3x3_s2_kernel  y[n,m,2Y+ky,2X+kx] += wgt(m,k,ky,kx) * x[n,k,Y,X]                          x: HxW -> y: (2H+1)x(2W+1)
//
// Reference: the `conv2d(stride=2)` that follows the FIR in the down-sampling path and the `conv_transpose2d(stride=2)` that
// precedes it in the up-sampling path of conv2d_resample (src/torch_utils/ops/conv2d_resample.py:113-137), plus each other's
// data gradients (conv2d_gradfix.py:100-118); MIOpen runs them as NHWC implicit GEMMs between layout transposes
// (profiles/r01_bench_step_kernel_stats_v3.csv: igemm bwd 47 + fwd 26 + transposes 31 ms of a 331 ms step).
//
// Both reuse the s1 kernel's scheme (conv3x3_kernel.h): MFMA rows = 32 output channels, columns = 32 pixels of the HxW grid,
// k = 16 input channels of one tap; x tile transposed into 16-B [8 channel] words in LDS; weights pre-arranged by
// conv3x3_prep_weights.  The (2W+1)-wide tensors have no 16-B aligned rows, so they are read / written with dword accesses
// (lanes over consecutive columns: fully coalesced).
//   * strided: the input tile is de-interleaved into even / odd column planes on its way into LDS, so that tap kx reads
//     plane kx & 1 at pixel + (kx >> 1): contiguous, conflict-free 16-B reads again.
//   * transposed: the nine taps scatter into four output parity classes (oy = 2Y + a, ox = 2X + b; 4 + 2 + 2 + 1 taps), each
//     with its own accumulator; the last output row (oy = 2H) and column (ox = 2W) are left to convT3x3_s2_edge_kernel.
#pragma once

#include "conv3x3_kernel.h"
#include "sgv_io16.h"

namespace sgv_conv {

// ------------------------------------------------------------------------------------------------------------------
// strided: tile 64 m x 8 output rows x 32 output px; wave = 2 rows x 64 m.
constexpr int S_ROWS = 8;
constexpr int S_RIN = 2 * S_ROWS + 1;          // 17 input rows
constexpr int S_PW = 33;                       // words per parity plane row (even plane holds 33 columns)
constexpr int S_XS_PLANE = S_RIN * 2 * S_PW;   // words per (hl, octet)
constexpr int S_XS_WORDS = 4 * S_XS_PLANE;
constexpr int S_LDS_BYTES = (S_XS_WORDS + WS_WORDS) * 16;
constexpr int S_PAIRS = 2 * S_RIN;             // (octet, row) pairs

struct s2_params {
    const float* x;
    const u32x4* wprep;
    float* y;
    int n, k, m, h, w;     // h, w: the SMALL grid (strided: output; transposed: input)
    int tiles, grid;
    int order;             // producer / consumer forms: 1 = a workgroup runs all m tiles of a spatial tile back to back
    const float* x_amax;   // TERMS = 4 (block-scaled fp16 split, sgv_split.h; producer / consumer forms): bounds of max |x| and max |weight|
    const float* w_amax;
};

__device__ __forceinline__ tile_pos decode_tile_s2(const s2_params& p, int tile, int rows) {
    const int mts = p.m / TM, segs = p.w / SEG, rbs = p.h / rows;
    tile_pos tp;
    tp.mt = tile % mts;
    int r = tile / mts;
    tp.x0 = (r % segs) * SEG;
    r /= segs;
    tp.y0 = (r % rbs) * rows;
    tp.n = r / rbs;
    return tp;
}

struct s_stage {
    float xv[9][8];   // up to 9 (octet, row) pairs per wave: 8 channels of column `lane`
    float xe[8];      // column 64 of pair t (threads < 34)
    u32x4 wv[9];
};

template <int TERMS>
__global__ __launch_bounds__(256, 1) void conv3x3_s2_kernel(s2_params p) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    u32x4* xs = lds;
    u32x4* ws = lds + S_XS_WORDS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l32 = lane & 31, g = lane >> 5;
    const int chunks = p.k / KC;
    const int hin = 2 * p.h + 1, win = 2 * p.w + 1;
    const size_t plane_in = (size_t)hin * win, plane_out = (size_t)p.h * p.w;

    auto load_chunk = [&](const tile_pos& tp, int c, s_stage& s) {
        const float* xb = p.x + ((size_t)tp.n * p.k + c * KC) * plane_in + (size_t)(2 * tp.y0) * win + 2 * tp.x0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int pi = wave + 4 * i;
            if (pi < S_PAIRS) {
                const float* q = xb + (size_t)(8 * (pi & 1)) * plane_in + (size_t)(pi >> 1) * win + lane;
#pragma unroll
                for (int j = 0; j < 8; j++) s.xv[i][j] = q[j * plane_in];
            }
        }
        if (t < S_PAIRS) {
            const float* q = xb + (size_t)(8 * (t & 1)) * plane_in + (size_t)(t >> 1) * win + 64;
#pragma unroll
            for (int j = 0; j < 8; j++) s.xe[j] = q[j * plane_in];
        }
        const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + c) * WS_WORDS + t;
#pragma unroll
        for (int j = 0; j < 9; j++) s.wv[j] = wq[j * 256];
    };
    auto store_chunk = [&](const s_stage& s) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int pi = wave + 4 * i;
            if (pi < S_PAIRS) {
                u32x4 hi, lo;
                split8(s.xv[i], hi, lo);
                const int pos = (((pi & 1) * S_RIN + (pi >> 1)) * 2 + (lane & 1)) * S_PW + (lane >> 1);
                xs[pos] = hi;
                if (TERMS > 1) xs[2 * S_XS_PLANE + pos] = lo;
            }
        }
        if (t < S_PAIRS) {
            u32x4 hi, lo;
            split8(s.xe, hi, lo);
            const int pos = (((t & 1) * S_RIN + (t >> 1)) * 2 + 0) * S_PW + 32;
            xs[pos] = hi;
            if (TERMS > 1) xs[2 * S_XS_PLANE + pos] = lo;
        }
#pragma unroll
        for (int j = 0; j < 9; j++) ws[t + j * 256] = s.wv[j];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][hf][e] = 0.f;

    int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    if (tile >= p.tiles) return;
    tile_pos tp = decode_tile_s2(p, tile, S_ROWS);
    int c = 0;
    {
        s_stage s;
        load_chunk(tp, 0, s);
        store_chunk(s);
        __syncthreads();
    }
    while (true) {
        int ntile = tile, nc = c + 1;
        if (nc == chunks) { nc = 0; ntile = tile + p.grid; }
        const bool more = ntile < p.tiles;
        tile_pos ntp = tp;
        if (more && nc == 0) ntp = decode_tile_s2(p, ntile, S_ROWS);
        s_stage s;
        if (more) load_chunk(ntp, nc, s);

#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
            u32x4 a[2][2];
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                a[hf][0] = ws[((0 * 9 + tap) * 2 + g) * TM + hf * 32 + l32];
                if (TERMS > 1) a[hf][1] = ws[((1 * 9 + tap) * 2 + g) * TM + hf * 32 + l32];
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int pos = ((g * S_RIN + 2 * (2 * wave + r) + ky) * 2 + (kx & 1)) * S_PW + l32 + (kx >> 1);
                const u32x4 b_hi = xs[pos];
                u32x4 b_lo;
                if (TERMS > 1) b_lo = xs[2 * S_XS_PLANE + pos];
                if (TERMS > 1) {
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][1]), __builtin_bit_cast(bf16x8, b_hi), acc[r][hf], 0, 0, 0);
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][0]), __builtin_bit_cast(bf16x8, b_lo), acc[r][hf], 0, 0, 0);
                }
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    acc[r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][0]), __builtin_bit_cast(bf16x8, b_hi), acc[r][hf], 0, 0, 0);
            }
        }

        if (c == chunks - 1) {
            float* yb = p.y + ((size_t)tp.n * p.m + tp.mt * TM) * plane_out + (size_t)(tp.y0 + 2 * wave) * p.w + tp.x0 + l32;
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int m = hf * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                        yb[(size_t)m * plane_out + (size_t)r * p.w] = acc[r][hf][e];
                        acc[r][hf][e] = 0.f;
                    }
        }
        if (!more) break;
        __syncthreads();
        store_chunk(s);
        __syncthreads();
        tile = ntile; c = nc; tp = ntp;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// transposed: tile 64 m x 8 input rows x 32 input px -> 16 x 64 outputs per channel.
constexpr int T_ROWS = 8;
constexpr int T_RIN = T_ROWS + 1;              // rows y0-1 .. y0+7
constexpr int T_PW = 33;                       // columns x0-1 .. x0+31
constexpr int T_XS_PLANE = T_RIN * T_PW;
constexpr int T_XS_WORDS = 4 * T_XS_PLANE;
constexpr int T_LDS_BYTES = (T_XS_WORDS + WS_WORDS) * 16;

struct t_stage {
    f32x4 xa[4];    // item t < 288: (channel quad of 16, row, px quad): 4 channels x 4 px
    float xh[4];    // item t < 36: (channel quad, row): column x0 - 1
    u32x4 wv[5];    // 2304 weight words over 512 threads
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 4 channel values of one pixel -> hi and lo halves (8 bytes each) of a [8 ch] LDS word
__device__ __forceinline__ void split4(const float* v, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const unsigned h = pack_bf16(v[2 * j], v[2 * j + 1]);
        const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
        hi[j] = h;
        lo[j] = pack_bf16(v[2 * j] - h0, v[2 * j + 1] - h1);
    }
}

// 8 waves (512 threads, 2 per SIMD, <= 256 registers each): a wave owns ONE input row x 4 parity classes x 64 m (128 accumulators).
template <int TERMS>
__global__ __launch_bounds__(512, 1) void convT3x3_s2_kernel(s2_params p) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    u32x4* xs = lds;
    u32x4* ws = lds + T_XS_WORDS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l32 = lane & 31, g = lane >> 5;
    const int chunks = p.k / KC;
    const int hout = 2 * p.h + 1, wout = 2 * p.w + 1;
    const size_t plane_in = (size_t)p.h * p.w, plane_out = (size_t)hout * wout;

    const int a_cq = t & 3, a_quad = (t >> 2) & 7, a_row = t >> 5;    // t < 288: channels 4*a_cq .. +3
    const int h_cq = t & 3, h_row = t >> 2;                             // t < 36

    auto load_chunk = [&](const tile_pos& tp, int c, t_stage& s) {
        const float* xb = p.x + ((size_t)tp.n * p.k + c * KC) * plane_in + tp.x0;
        if (t < 4 * T_RIN * 8) {
            const int gy = tp.y0 - 1 + a_row;
            const bool ok = gy >= 0;
            const float* q = xb + (size_t)(4 * a_cq) * plane_in + (size_t)gy * p.w + 4 * a_quad;
#pragma unroll
            for (int j = 0; j < 4; j++) s.xa[j] = ok ? *(const f32x4*)(q + j * plane_in) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (t < 4 * T_RIN) {
            const int gy = tp.y0 - 1 + h_row;
            const bool ok = gy >= 0 && tp.x0 > 0;
            const float* q = xb + (size_t)(4 * h_cq) * plane_in + (size_t)gy * p.w - 1;
#pragma unroll
            for (int j = 0; j < 4; j++) s.xh[j] = ok ? q[j * plane_in] : 0.f;
        }
        const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + c) * WS_WORDS + t;
#pragma unroll
        for (int j = 0; j < 5; j++)
            if (t + j * 512 < WS_WORDS) s.wv[j] = wq[j * 512];
    };
    auto store_chunk = [&](const t_stage& s) {
        u32x2* xs2 = (u32x2*)xs;   // half words: [word][channel quad & 1]
        if (t < 4 * T_RIN * 8) {
            const int base = ((a_cq >> 1) * T_RIN + a_row) * T_PW + 1 + 4 * a_quad;
#pragma unroll
            for (int px = 0; px < 4; px++) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = s.xa[j][px];
                u32x2 hi, lo;
                split4(v, hi, lo);
                xs2[2 * (base + px) + (a_cq & 1)] = hi;
                if (TERMS > 1) xs2[2 * (2 * T_XS_PLANE + base + px) + (a_cq & 1)] = lo;
            }
        }
        if (t < 4 * T_RIN) {
            u32x2 hi, lo;
            split4(s.xh, hi, lo);
            const int pos = ((h_cq >> 1) * T_RIN + h_row) * T_PW;
            xs2[2 * pos + (h_cq & 1)] = hi;
            if (TERMS > 1) xs2[2 * (2 * T_XS_PLANE + pos) + (h_cq & 1)] = lo;
        }
#pragma unroll
        for (int j = 0; j < 5; j++)
            if (t + j * 512 < WS_WORDS) ws[t + j * 512] = s.wv[j];
    };

    f32x16 acc[4][2];   // [class a*2+b][m half]
#pragma unroll
    for (int cl = 0; cl < 4; cl++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[cl][hf][e] = 0.f;

    int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    if (tile >= p.tiles) return;
    tile_pos tp = decode_tile_s2(p, tile, T_ROWS);
    int c = 0;
    {
        t_stage s;
        load_chunk(tp, 0, s);
        store_chunk(s);
        __syncthreads();
    }
    while (true) {
        int ntile = tile, nc = c + 1;
        if (nc == chunks) { nc = 0; ntile = tile + p.grid; }
        const bool more = ntile < p.tiles;
        tile_pos ntp = tp;
        if (more && nc == 0) ntp = decode_tile_s2(p, ntile, T_ROWS);
        t_stage s;
        if (more) load_chunk(ntp, nc, s);

#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
            const int cl = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
            const int dy = ky == 2 ? 1 : 0, dx = kx == 2 ? 1 : 0;   // tap 2 reaches back to the previous input row / column
            const int pos = (g * T_RIN + wave + 1 - dy) * T_PW + l32 + 1 - dx;
            const u32x4 b_hi = xs[pos];
            u32x4 b_lo;
            if (TERMS > 1) b_lo = xs[2 * T_XS_PLANE + pos];
            u32x4 a[2][2];
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                a[hf][0] = ws[((0 * 9 + tap) * 2 + g) * TM + hf * 32 + l32];
                if (TERMS > 1) a[hf][1] = ws[((1 * 9 + tap) * 2 + g) * TM + hf * 32 + l32];
            }
            if (TERMS > 1) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    acc[cl][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][1]), __builtin_bit_cast(bf16x8, b_hi), acc[cl][hf], 0, 0, 0);
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    acc[cl][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][0]), __builtin_bit_cast(bf16x8, b_lo), acc[cl][hf], 0, 0, 0);
            }
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
                acc[cl][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][0]), __builtin_bit_cast(bf16x8, b_hi), acc[cl][hf], 0, 0, 0);
        }

        if (c == chunks - 1) {
            float* yb = p.y + ((size_t)tp.n * p.m + tp.mt * TM) * plane_out + (size_t)(2 * (tp.y0 + wave)) * wout + 2 * (tp.x0 + l32);
#pragma unroll
            for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int m = hf * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                        float* q = yb + (size_t)m * plane_out + (size_t)a2 * wout;
                        q[0] = acc[a2 * 2 + 0][hf][e];
                        q[1] = acc[a2 * 2 + 1][hf][e];
                        acc[a2 * 2 + 0][hf][e] = 0.f;
                        acc[a2 * 2 + 1][hf][e] = 0.f;
                    }
        }
        if (!more) break;
        __syncthreads();
        store_chunk(s);
        __syncthreads();
        tile = ntile; c = nc; tp = ntp;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Last output row (oy = 2H: taps ky = 2 of input row H-1) and last output column (ox = 2W: taps kx = 2 of input column W-1)
// of the transposed convolution.  0.4 % of the flops, done in plain fp32 FMAs in two steps so that every access is coalesced:
//   convT3x3_s2_edge_gather   edge[0][n][k][0..W) = x[n,k,H-1,:]   edge[1][n][k][0..H) = x[n,k,:,W-1]
//   convT3x3_s2_edge_kernel   1-D transposed convolutions of those lines with w[k][m][2][:] resp. w[k][m][:][2]
__global__ __launch_bounds__(256) void convT3x3_s2_edge_gather(const float* x, const float* w, float* edge, int n, int k, int m, int h, int wd) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nk = (size_t)n * k;
    const size_t n_row = nk * wd, n_col = nk * h, n_w = (size_t)2 * k * 3 * m;
    if (idx < n_row) {
        const size_t c = idx / wd; const int i = idx % wd;
        edge[idx] = x[(c * h + (h - 1)) * wd + i];
    } else if (idx < n_row + n_col) {
        const size_t j = idx - n_row;
        const size_t c = j / h; const int i = j % h;
        edge[idx] = x[(c * h + i) * wd + (wd - 1)];
    } else if (idx < n_row + n_col + n_w) {
        // strip weights we[strip][k][tap][m]: row strip -> w[k][m][2][tap], column strip -> w[k][m][tap][2]
        size_t j = idx - n_row - n_col;
        const int mm = j % m; j /= m;
        const int tap = j % 3; j /= 3;
        const int kk = j % k; const int strip = j / k;
        edge[idx] = w[((size_t)kk * m + mm) * 9 + (strip == 0 ? 6 + tap : 3 * tap + 2)];
    }
}

constexpr int EDGE_MC = 16;   // output channels per thread

__host__ __device__ inline size_t convT3x3_s2_edge_floats(int n, int k, int m, int h, int wd) { return (size_t)n * k * (h + wd) + (size_t)2 * k * 3 * m; }

// grid = (ceil(max(2W+1, 2H) / 128), n * (m / 16), 2 strips), 128 threads.
__global__ __launch_bounds__(128) void convT3x3_s2_edge_kernel(const float* edge, float* y, int n, int k, int m, int h, int wd) {
    const int strip = blockIdx.z;                     // 0: bottom row, 1: right column
    const int ls = strip == 0 ? wd : h;               // source line length
    const int lo = strip == 0 ? 2 * wd + 1 : 2 * h;   // outputs (the column strip leaves the corner to the row strip)
    if ((int)blockIdx.x * 128 >= lo) return;
    const int pos = blockIdx.x * 128 + threadIdx.x;
    const int mchunks = m / EDGE_MC;
    const int nn = blockIdx.y / mchunks, m0 = (blockIdx.y % mchunks) * EDGE_MC;
    const int hout = 2 * h + 1, wout = 2 * wd + 1;
    const float* src = edge + (strip == 0 ? 0 : (size_t)n * k * wd) + (size_t)nn * k * ls;
    const float* we = edge + (size_t)n * k * (h + wd) + (size_t)strip * k * 3 * m + m0;   // [k][tap][m]
    const bool odd = pos & 1;
    const int ia = pos >> 1, ib = (pos >> 1) - 1;     // sources of tap (odd ? 1 : 0) and of tap 2
    const bool va = pos < lo && ia < ls, vb = pos < lo && !odd && ib >= 0;
    float accv[EDGE_MC];
#pragma unroll
    for (int j = 0; j < EDGE_MC; j++) accv[j] = 0.f;
    // k is a multiple of 16: 8 source pairs are fetched ahead of their FMAs (the loop is latency-bound otherwise)
    for (int k0 = 0; k0 < k; k0 += 8) {
        float xa[8], xb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            xa[u] = va ? src[(size_t)(k0 + u) * ls + ia] : 0.f;
            xb[u] = vb ? src[(size_t)(k0 + u) * ls + ib] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float* wp = we + (size_t)(k0 + u) * 3 * m;   // uniform over the workgroup, 16 consecutive floats per tap -> scalar loads
#pragma unroll
            for (int j = 0; j < EDGE_MC; j++) {
                const float w0 = wp[j], w1 = wp[m + j], w2 = wp[2 * m + j];
                accv[j] = __builtin_fmaf(odd ? w1 : w0, xa[u], accv[j]);
                accv[j] = __builtin_fmaf(w2, xb[u], accv[j]);
            }
        }
    }
    if (pos >= lo) return;
    float* yb = y + ((size_t)nn * m + m0) * hout * wout + (strip == 0 ? (size_t)(hout - 1) * wout + pos : (size_t)pos * wout + (wout - 1));
#pragma unroll
    for (int j = 0; j < EDGE_MC; j++) yb[(size_t)j * hout * wout] = accv[j];
}

// convT3x3_s2_edge_prep + convT3x3_s2_edge_mfma: both strips in one launch on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32; plain fp32 products).
// A wave owns 32 output channels x 32 source positions i of one (sample, strip) and keeps two accumulator tiles:
//   even outputs  2i   = sum_k wt[0](m,k) * src(k,i) + wt[2](m,k) * src(k,i-1)
//   odd outputs   2i+1 = sum_k wt[1](m,k) * src(k,i)
// with (row strip, oy = 2H) src(k,i) = x[n,k,H-1,i], wt[t] = w[k][m][2][t] and (column strip, ox = 2W, without the corner) src(k,i) = x[n,k,i,W-1],
// wt[t] = w[k][m][t][2].  MFMA operands are one float per lane: A lane (m = lane & 31, k = k0 + (lane >> 5)), B lane (i = lane & 31, same k).
// Reading w (stride 9 floats over m) and the column x[:, :, :, W-1] (stride W) in place makes every load touch 10-32 cache lines and the kernel
// address-unit bound (0.25 ms per layer); the prep kernel lays both out once:  edge = we[strip][tap][k][m] (2*3*K*M floats), col[n][k][H].
__host__ __device__ inline size_t convT3x3_s2_edge_we_floats(int k, int m) { return (size_t)2 * 3 * k * m; }

// IO != 0 (16-bit x, sgv_io16.h): the weights are rounded to the tensor's format (bf16 / fp16) like the main kernel's operands, and the last input ROW is gathered as fp32 too
// (row[n][k][W] behind col[n][k][H]) so that the strip kernel reads fp32 lines for both strips.
template <int IO>
__global__ __launch_bounds__(256) void convT3x3_s2_edge_prep(const void* x, const float* w, float* edge, int n, int k, int m, int h, int wd) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_w = convT3x3_s2_edge_we_floats(k, m), n_col = (size_t)n * k * h, n_row = IO ? (size_t)n * k * wd : 0;
    auto ld = [&](size_t e) -> float {
        if constexpr (IO == 0) return ((const float*)x)[e];
        else if constexpr (IO == 1) return __builtin_bit_cast(float, (unsigned)((const uint16_t*)x)[e] << 16);
        else return (float)((const _Float16*)x)[e];      // fp16 tensors are multiplied as they are (operand_format of sgv_split.h)
    };
    if (idx < n_w) {
        size_t j = idx;
        const int mm = j % m; j /= m;
        const int kk = j % k; j /= k;
        const int tap = j % 3; const int strip = j / 3;
        float v = w[((size_t)kk * m + mm) * 9 + (strip == 0 ? 6 + tap : 3 * tap + 2)];
        if (IO == 1) v = __builtin_bit_cast(float, pack_bf16(v, 0.f) << 16);
        if (IO == 2) v = (float)(_Float16)v;
        edge[idx] = v;
    } else if (idx < n_w + n_col) {
        const size_t j = idx - n_w;
        const size_t c = j / h; const int i = j % h;
        edge[idx] = ld((c * h + i) * wd + (wd - 1));
    } else if (idx < n_w + n_col + n_row) {
        const size_t j = idx - n_w - n_col;
        const size_t c = j / wd; const int i = j % wd;
        edge[idx] = ld((c * h + (h - 1)) * wd + i);
    }
}

typedef float f32x16e __attribute__((ext_vector_type(16)));

// grid = (ceil((max(H, W) + 1) / 32), n * (m / 32), 2 strips), 64 * NW threads: the NW waves of a workgroup each take k / NW input channels of the same
// 32 x 32 output block and are summed through LDS by wave 0.  (A wave's k loop is a serial chain of load -> MFMA steps, k / 8 of them, and the grid has
// only a few waves per SIMD: with one wave per block the 512-channel layers took 0.23 ms per call for 0.4 % of the layer's flops -- latency, not work.)
template <int IO, int NW = 1>
__global__ __launch_bounds__(64 * NW) void convT3x3_s2_edge_mfma(const float* x, const float* edge, void* y, int n, int k, int m, int h, int wd) {
    const int strip = blockIdx.z;
    const int ls = strip == 0 ? wd : h;               // source line length
    const int lo = strip == 0 ? 2 * wd + 1 : 2 * h;   // outputs of the strip
    const int i0 = blockIdx.x * 32;
    if (2 * i0 >= lo) return;
    const int mts = m / 32;
    const int nn = blockIdx.y / mts, m0 = (blockIdx.y % mts) * 32;
    const int lane = threadIdx.x & 63, l32 = lane & 31, g = lane >> 5;
    const int kw = NW > 1 ? (int)(threadIdx.x >> 6) : 0, kper = k / NW, kb = kw * kper;     // this wave's channels: kb .. kb + kper (host: kper % 8 == 0)
    const int i = i0 + l32;
    const bool vb = i < ls, vm = i >= 1 && i - 1 < ls;
    const size_t plane = (size_t)h * wd;
    // src(k, i) = xs[k * kstep + i]: the row strip reads x[n, k, H-1, :] in place, the column strip its gathered copy
    // (16-bit tensors: the row strip reads its gathered fp32 copy as well, x is not touched)
    const float* xs = strip == 0 ? (IO ? edge + convT3x3_s2_edge_we_floats(k, m) + (size_t)n * k * h + (size_t)nn * k * wd : x + (size_t)nn * k * plane + (size_t)(h - 1) * wd)
                                 : edge + convT3x3_s2_edge_we_floats(k, m) + (size_t)nn * k * h;
    const size_t kstep = strip == 0 ? (IO ? (size_t)wd : plane) : (size_t)h;
    const float* pb = xs + (vb ? i : 0) + (size_t)(kb + g) * kstep;
    const float* pm = xs + (vm ? i - 1 : 0) + (size_t)(kb + g) * kstep;
    const size_t tk = (size_t)k * m;                  // floats per tap
    const float* pw = edge + (size_t)strip * 3 * tk + (size_t)(kb + g) * m + m0 + l32;
    const size_t wk = (size_t)2 * m, xk = 2 * kstep;

    f32x16e acc_e, acc_o;
#pragma unroll
    for (int e = 0; e < 16; e++) { acc_e[e] = 0.f; acc_o[e] = 0.f; }
    constexpr int U = 4;   // k pairs in flight (k is a multiple of 16)
    for (int k0 = 0; k0 < kper; k0 += 2 * U) {
        float a0[U], a1[U], a2[U], b[U], bm[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            a0[u] = pw[u * wk];
            a1[u] = pw[u * wk + tk];
            a2[u] = pw[u * wk + 2 * tk];
            b[u] = vb ? pb[u * xk] : 0.f;
            bm[u] = vm ? pm[u * xk] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc_e = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b[u], acc_e, 0, 0, 0);
            acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b[u], acc_o, 0, 0, 0);
            acc_e = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[u], bm[u], acc_e, 0, 0, 0);
        }
        pw += U * wk; pb += U * xk; pm += U * xk;
    }
    if (NW > 1) {
        __shared__ float red[NW > 1 ? NW - 1 : 1][32][64];     // [wave - 1][accumulator word][lane]: conflict-free
        if (kw > 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) { red[kw - 1][e][lane] = acc_e[e]; red[kw - 1][16 + e][lane] = acc_o[e]; }
        }
        __syncthreads();
        if (kw > 0) return;
#pragma unroll
        for (int w2 = 0; w2 < NW - 1; w2++)
#pragma unroll
            for (int e = 0; e < 16; e++) { acc_e[e] += red[w2][e][lane]; acc_o[e] += red[w2][16 + e][lane]; }
    }
    const int hout = 2 * h + 1, wout = 2 * wd + 1;
    const int pe = 2 * i, po = 2 * i + 1;
    // element at position `pos` of the strip: row strip y[.., 2H, pos], column strip y[.., pos, 2W]
    const size_t yb = ((size_t)nn * m + m0) * hout * wout + (strip == 0 ? (size_t)(hout - 1) * wout : (size_t)(wout - 1));
    const size_t ystep = strip == 0 ? 1 : wout;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int mm = (e & 3) + 8 * (e >> 2) + 4 * g;
        const size_t q = yb + (size_t)mm * hout * wout;
        if (pe < lo) sgv_io::out_store<IO>(y, q + (size_t)pe * ystep, acc_e[e]);
        if (po < lo) sgv_io::out_store<IO>(y, q + (size_t)po * ystep, acc_o[e]);
    }
}

}  // namespace sgv_conv
