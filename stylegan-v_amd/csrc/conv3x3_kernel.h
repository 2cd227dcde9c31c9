// 3x3 / stride 1 / pad 1 convolution (forward, and data gradient with re-indexed weights) on NCHW fp32 tensors, on the gfx950
// matrix cores with bf16x3 fp32 emulation (see wrw_kernel.h for the arithmetic).
//
//     y[n,m,Y,X] = sum_{k,ky,kx} wgt(m,k,ky,kx) * x[n,k,Y+ky-1,X+kx-1]
//
// Reference: the `conv2d` of SynthesisLayer / Conv2dLayer (src/training/networks.py:58-62 via conv2d_resample.py:40-54) and its
// data gradient (conv2d_gradfix.py:100-118 `conv_transpose2d`), cuDNN in the reference, MIOpen's fp32 Winograd here
// (profiles/r01_bench_step_kernel_stats_v2.csv: 37 % of the train step at ~100 TFLOP/s effective).
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16): rows = 32 output channels, columns = 32 consecutive pixels of one image row,
// k = 16 input channels of one tap.  NCHW makes k the strided index, so the x tile is transposed on its way into LDS: a thread
// loads 8 channels x 4 pixels (eight 16-B loads; every 128-B line is consumed whole by one instruction) and writes, per pixel,
// the 8 channels as one 16-B bf16 vector.  LDS layout x: [hi/lo][channel octet][row][pixel][8 ch] -> the B operand of tap
// (ky,kx) for pixel p is the 16-B word at [row + ky][p + kx]: a conflict-free ds_read_b128, the nine taps are nine addresses.
// Weights are pre-arranged (prep kernel below) as [m tile][k chunk][hi/lo][tap][octet][64 m][8 k] bf16 and copied verbatim.
//
// Workgroup = 4 waves, output tile 64 m x 16 rows x 32 px; a wave owns 4 rows x 64 m = 8 accumulator tiles (128 registers).
// Per 16-channel chunk a wave issues 9 taps x (4 A reads + 4 x (2 B reads + 6 MFMAs)) = 216 MFMAs; the next chunk's global
// loads (x and weights) are in flight meanwhile, held in registers, split into hi/lo and written to LDS between two barriers.
// One workgroup per CU (75 KB LDS, ~300 registers), persistent over (sample, row block, column segment, m tile).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "sgv_split.h"

namespace sgv_conv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64;            // output channels per workgroup
constexpr int TROWS = 16;         // output rows per workgroup in the default configuration (RW = 4 rows per wave)
constexpr int SEG = 32;           // output pixels per row
constexpr int KC = 16;            // input channels per chunk (= MFMA k)
constexpr int RIN = TROWS + 2;    // input rows per tile
constexpr int PIN = SEG + 2;      // input pixels per row
constexpr int XS_PLANE = RIN * PIN;             // 16-B words per (hi/lo, octet) plane
constexpr int XS_WORDS = 2 * 2 * XS_PLANE;      // u32x4 words
constexpr int WS_WORDS = 2 * 9 * 2 * TM;        // [hl][tap][octet][64 m] u32x4 words
constexpr int LDS_BYTES = (XS_WORDS + WS_WORDS) * 16;
constexpr int lds_bytes_rw(int rw) { return (4 * (4 * rw + 2) * PIN + WS_WORDS) * 16; }

struct conv_params {
    const float* x;        // [n, k, h, w]   (conv3x3_ws_kernel with IO != 0: 16-bit elements behind the same pointers)
    const u32x4* wprep;    // [m tiles][k chunks][hl][tap][octet][64][8 bf16]
    float* y;              // [n, m, h, w]
    int n, k, m, h, w;
    int tiles;             // n * (h/16) * (w/32) * (m/64)
    int grid;              // persistent workgroups
    // TERMS = 4 (block-scaled fp16 split, sgv_split.h): device pointers to upper bounds of max |x| (optionally a second factor: x * styles) and of
    // max |weight| (written by sgv_absmax_kernel in front of the weight preparation); NULL otherwise
    const float* x_amax;
    const float* x_amax2;
    const float* w_amax;
    int ksplit;            // conv3x3_small_kernel: the input channels of a tile are shared out over `ksplit` workgroups (tiles counts them), which ADD their partial
                           // sums into a zeroed y with no-return fp32 atomics; 0 / 1: one workgroup per tile, plain stores
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// 8 channel values of one pixel -> hi and lo bf16x8
__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const unsigned h = pack_bf16(v[2 * j], v[2 * j + 1]);
        const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
        hi[j] = h;
        lo[j] = pack_bf16(v[2 * j] - h0, v[2 * j + 1] - h1);
    }
}

struct stage_regs {
    f32x4 xa[8];     // item t:        8 channels x 4 pixels
    f32x4 xb[8];     // item t + 256 (threads < 32)
    float xh[8];     // halo item (threads < 72): 8 channels of one (row, side)
    u32x4 wv[9];     // 9 x 16 B of the 36-KiB weight chunk
};

struct tile_pos { int n, y0, x0, mt; };

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Consecutive tile indices differ in the output-channel
// tile and read the same x tile, so workgroup i takes the i-th entry of a per-XCD contiguous range: neighbours in tile order share an L2.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblocks) {
    return (nblocks & 7) == 0 ? (bid & 7) * (nblocks >> 3) + (bid >> 3) : bid;
}

__device__ __forceinline__ tile_pos decode_tile(const conv_params& p, int tile, int trows = TROWS) {
    const int mts = (p.m + TM - 1) / TM, segs = p.w / SEG, rbs = p.h / trows;   // (a last m tile may be half full: m % 32 == 0, conv3x3_ws_kernel only)
    tile_pos tp;
    tp.mt = tile % mts;
    int r = tile / mts;
    tp.x0 = (r % segs) * SEG;
    r /= segs;
    tp.y0 = (r % rbs) * trows;
    tp.n = r / rbs;
    return tp;
}

// ABL (tools/conv_lab.hip only): 1 = skip the weight copy, 2 = skip the x split + LDS fill, 3 = skip the global loads, 4 = skip the MFMAs.
// RW = output rows per wave: 4 (tile 16 rows, one workgroup per CU) or 2 (tile 8 rows, <= 256 registers and 59 KiB LDS: two workgroups per
// CU, so that one's LDS fill overlaps the other's MFMAs).
template <int TERMS, int ABL = 0, int RW = 4>
__global__ __launch_bounds__(256, RW == 4 ? 1 : 2) void conv3x3_kernel(conv_params p) {
    constexpr int TROWS = 4 * RW, RIN = TROWS + 2, XS_PLANE = RIN * PIN, XS_WORDS = 4 * XS_PLANE, ITEMS = 16 * RIN;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    u32x4* xs = lds;                // [hl][octet][row][px]
    u32x4* ws = lds + XS_WORDS;     // [hl][tap][octet][m]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, g = lane >> 5;
    const int chunks = p.k / KC;
    const size_t plane = (size_t)p.h * p.w;

    // loader roles
    const int a_oct = t & 1, a_quad = (t >> 1) & 7, a_row = t >> 4;            // item t (< ITEMS): rows 0..15
    const int b_row = 16 + (t >> 4);                                           // item t + 256 (< ITEMS): rows 16, 17
    const int h_oct = t & 1, h_side = (t >> 1) & 1, h_row = t >> 2;            // halo item (t < 4 * RIN)

    auto load_chunk = [&](const tile_pos& tp, int c, stage_regs& s) {
        const float* xb = p.x + ((size_t)tp.n * p.k + c * KC) * plane + tp.x0;
        if (t < ITEMS) {
            const int gy = tp.y0 - 1 + a_row;
            const bool ok = gy >= 0 && gy < p.h;
            const float* q = xb + (size_t)(8 * a_oct) * plane + (size_t)gy * p.w + 4 * a_quad;
#pragma unroll
            for (int j = 0; j < 8; j++) s.xa[j] = ok ? *(const f32x4*)(q + j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (t + 256 < ITEMS) {
            const int gy = tp.y0 - 1 + b_row;
            const bool ok = gy < p.h;
            const float* q = xb + (size_t)(8 * a_oct) * plane + (size_t)gy * p.w + 4 * a_quad;
#pragma unroll
            for (int j = 0; j < 8; j++) s.xb[j] = ok ? *(const f32x4*)(q + j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (t < 4 * RIN) {
            const int gy = tp.y0 - 1 + h_row;
            const int gx = h_side ? tp.x0 + SEG : tp.x0 - 1;
            const bool ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
            const float* q = xb + (size_t)(8 * h_oct) * plane + (size_t)gy * p.w + (gx - tp.x0);
#pragma unroll
            for (int j = 0; j < 8; j++) s.xh[j] = ok ? q[j * plane] : 0.f;
        }
        const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + c) * WS_WORDS + t;
#pragma unroll
        for (int j = 0; j < 9; j++) s.wv[j] = wq[j * 256];
    };

    auto store_chunk = [&](const stage_regs& s) {
        if (ABL != 2 && t < ITEMS) {
            const int base = (a_oct * RIN + a_row) * PIN + 1 + 4 * a_quad;
#pragma unroll
            for (int px = 0; px < 4; px++) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = s.xa[j][px];
                u32x4 hi, lo;
                split8(v, hi, lo);
                xs[base + px] = hi;
                if (TERMS > 1) xs[2 * XS_PLANE + base + px] = lo;
            }
        }
        if (t + 256 < ITEMS && ABL != 2) {
            const int base = (a_oct * RIN + b_row) * PIN + 1 + 4 * a_quad;
#pragma unroll
            for (int px = 0; px < 4; px++) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = s.xb[j][px];
                u32x4 hi, lo;
                split8(v, hi, lo);
                xs[base + px] = hi;
                if (TERMS > 1) xs[2 * XS_PLANE + base + px] = lo;
            }
        }
        if (t < 4 * RIN && ABL != 2) {
            u32x4 hi, lo;
            split8(s.xh, hi, lo);
            const int pos = (h_oct * RIN + h_row) * PIN + (h_side ? SEG + 1 : 0);
            xs[pos] = hi;
            if (TERMS > 1) xs[2 * XS_PLANE + pos] = lo;
        }
        if (ABL != 1) {
#pragma unroll
            for (int j = 0; j < 9; j++) ws[t + j * 256] = s.wv[j];
        }
    };

    f32x16 acc[RW][2];
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][hf][e] = 0.f;

    int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    if (tile >= p.tiles) return;
    tile_pos tp = decode_tile(p, tile, TROWS);
    int c = 0;
    {
        stage_regs s;
        load_chunk(tp, 0, s);
        store_chunk(s);
        __syncthreads();
    }

    while (true) {
        // what comes after (tile, c)
        int ntile = tile, nc = c + 1;
        if (nc == chunks) { nc = 0; ntile = tile + p.grid; }
        const bool more = ntile < p.tiles;
        tile_pos ntp = tp;
        if (more && nc == 0) ntp = decode_tile(p, ntile, TROWS);
        stage_regs s;
        if (more && ABL != 3) load_chunk(ntp, nc, s);

        // ---- 216 MFMAs on chunk c ----
#pragma unroll
        for (int tap = 0; tap < (ABL == 4 ? 0 : 9); tap++) {
            const int ky = tap / 3, kx = tap % 3;
            u32x4 a[2][2];   // [half][hl]
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                a[hf][0] = ws[((0 * 9 + tap) * 2 + g) * TM + hf * 32 + l32];
                if (TERMS > 1) a[hf][1] = ws[((1 * 9 + tap) * 2 + g) * TM + hf * 32 + l32];
            }
            // B operands of the wave's 4 rows first, then three passes over the 8 accumulators: an MFMA never waits for the
            // accumulator written by the one just before it.
            u32x4 b_hi[RW], b_lo[RW];
#pragma unroll
            for (int r = 0; r < RW; r++) {
                const int pos = (g * RIN + RW * wave + r + ky) * PIN + l32 + kx;
                b_hi[r] = xs[pos];
                if (TERMS > 1) b_lo[r] = xs[2 * XS_PLANE + pos];
            }
            if (TERMS > 1) {
#pragma unroll
                for (int r = 0; r < RW; r++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][1]), __builtin_bit_cast(bf16x8, b_hi[r]), acc[r][hf], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RW; r++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][0]), __builtin_bit_cast(bf16x8, b_lo[r]), acc[r][hf], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < RW; r++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    acc[r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][0]), __builtin_bit_cast(bf16x8, b_hi[r]), acc[r][hf], 0, 0, 0);
        }

        if (c == chunks - 1) {
            // C layout: col (pixel) = lane & 31, row (m) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5): 128-B contiguous stores
            float* yb = p.y + ((size_t)tp.n * p.m + tp.mt * TM) * plane + (size_t)(tp.y0 + RW * wave) * p.w + tp.x0 + l32;
#pragma unroll
            for (int r = 0; r < RW; r++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int m = hf * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                        yb[(size_t)m * plane + (size_t)r * p.w] = acc[r][hf][e];
                        acc[r][hf][e] = 0.f;
                    }
        }
        if (!more) break;
        __syncthreads();      // every wave is done reading chunk c
        if (ABL != 3) store_chunk(s);
        __syncthreads();
        tile = ntile; c = nc; tp = ntp;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Small square images (SW x SW, SW = 16, 8 or 4: the 16^2 / 8^2 / 4^2 blocks).  Same MFMA mapping, but the 512 pixels of a workgroup
// tile are S = 512 / SW^2 WHOLE images (2 at 16^2, 8 at 8^2, 32 at 4^2): a 32-pixel MFMA column block is 32 / SW consecutive image rows.
// The batch need not be a multiple of S (round 5): samples beyond it are zero images that are never stored.
// Every halo position of the LDS image (row -1, row SW, column -1, column SW of each sample) is zero padding: it is cleared
// once at kernel start and never written again; per 16-channel chunk each thread loads exactly one (8 channels x 4 pixels) item.
template <int SW> struct small_cfg {
    static constexpr int S = 512 / (SW * SW);        // samples per tile
    static constexpr int RINS = S * (SW + 2);        // LDS rows
    static constexpr int PINS = SW + 2;              // LDS columns
    static constexpr int PLANE = RINS * PINS;        // words per (hl, octet)
    static constexpr int XS = 4 * PLANE;
    static constexpr int QPR = SW / 4;               // 4-pixel quads per image row
    static constexpr int LDS = (XS + WS_WORDS) * 16;
};

struct small_stage { f32x4 xa[8]; u32x4 wv[9]; };

template <int TERMS, int SW>
__global__ __launch_bounds__(256, 1) void conv3x3_small_kernel(conv_params p) {
    typedef small_cfg<SW> C;
    const int ex = operand_exponent<TERMS>(p.x_amax, p.x_amax2), ew = operand_exponent<TERMS>(p.w_amax);
    const float xS = split_scale(ex);
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    u32x4* xs = lds;
    u32x4* ws = lds + C::XS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l32 = lane & 31, g = lane >> 5;
    const int chunks = p.k / KC, mts = p.m / TM;
    constexpr int PLANE_PX = SW * SW;
    const size_t plane = PLANE_PX;

    for (int i = t; i < C::XS; i += 256) xs[i] = u32x4{0u, 0u, 0u, 0u};   // zero padding (and everything else, once)

    // loader role: 8 channels (octet) x 4 pixels of image row R of the tile (R = sample * SW + y)
    const int a_oct = t & 1, a_quad = (t >> 1) % C::QPR, a_R = (t >> 1) / C::QPR;
    const int a_s = a_R / SW, a_y = a_R % SW;
    const int a_pos = (a_oct * C::RINS + a_s * (SW + 2) + a_y + 1) * C::PINS + 1 + 4 * a_quad;

    auto load_chunk = [&](int n0, int mt, int c, small_stage& s) {
        const bool live = n0 + a_s < p.n;      // (a last tile may hold fewer than S samples)
        const float* q = p.x + ((size_t)(live ? n0 + a_s : n0) * p.k + c * KC + 8 * a_oct) * plane + a_y * SW + 4 * a_quad;
#pragma unroll
        for (int j = 0; j < 8; j++) s.xa[j] = live ? *(const f32x4*)(q + j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
        const u32x4* wq = p.wprep + ((size_t)mt * chunks + c) * WS_WORDS + t;
#pragma unroll
        for (int j = 0; j < 9; j++) s.wv[j] = wq[j * 256];
    };
    auto store_chunk = [&](const small_stage& s) {
#pragma unroll
        for (int px = 0; px < 4; px++) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = s.xa[j][px];
            u32x4 hi, lo;
            split8t<TERMS>(v, xS, hi, lo);
            xs[a_pos + px] = hi;
            if (TERMS > 1) xs[2 * C::PLANE + a_pos + px] = lo;
        }
#pragma unroll
        for (int j = 0; j < 9; j++) ws[t + j * 256] = s.wv[j];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][hf][e] = 0.f;

    // B-operand position of this lane in each of the wave's 4 column blocks (pixels 32 * slot + l32 of the tile)
    int bpos[4], ypix[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int q = 32 * (4 * wave + r) + l32;
        const int R = q / SW, px = q % SW, sidx = R / SW, yy = R % SW;
        bpos[r] = (g * C::RINS + sidx * (SW + 2) + yy) * C::PINS + px;
        ypix[r] = q;   // pixel index inside the tile: sample = q / SW^2, offset inside the sample plane = q % SW^2
    }

    // Few tiles (a rank's 24 ... 96 frames at 8^2: 24 ... 96 tiles for 256 CUs): a tile's K loop is shared out over `ksplit` workgroups -- unit u is K slice
    // u % ksplit of tile u / ksplit -- and the partial sums meet in y through atomics (the launcher zeroes y first).
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1, cps = chunks / ksplit;
    int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    if (tile >= p.tiles) return;
    int mt = (tile / ksplit) % mts, n0 = ((tile / ksplit) / mts) * C::S;
    int c = (tile % ksplit) * cps, c_end = c + cps;
    __syncthreads();   // zero fill complete
    {
        small_stage s;
        load_chunk(n0, mt, c, s);
        store_chunk(s);
        __syncthreads();
    }
    while (true) {
        int ntile = tile, nc = c + 1;
        if (nc == c_end) { ntile = tile + p.grid; nc = (ntile % ksplit) * cps; }
        const bool more = ntile < p.tiles;
        const int nmt = (ntile / ksplit) % mts, nn0 = ((ntile / ksplit) / mts) * C::S;
        small_stage s;
        if (more) load_chunk(nn0, nmt, nc, s);

#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
            u32x4 a[2][2];
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                a[hf][0] = ws[((0 * 9 + tap) * 2 + g) * TM + hf * 32 + l32];
                if (TERMS > 1) a[hf][1] = ws[((1 * 9 + tap) * 2 + g) * TM + hf * 32 + l32];
            }
            u32x4 b_hi[4], b_lo[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pos = bpos[r] + ky * C::PINS + kx;
                b_hi[r] = xs[pos];
                if (TERMS > 1) b_lo[r] = xs[2 * C::PLANE + pos];
            }
            if (TERMS > 1) {
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[r][hf] = mma16<TERMS>(a[hf][1], b_hi[r], acc[r][hf]);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[r][hf] = mma16<TERMS>(a[hf][0], b_lo[r], acc[r][hf]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    acc[r][hf] = mma16<TERMS>(a[hf][0], b_hi[r], acc[r][hf]);
        }

        if (c == c_end - 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool live = n0 + ypix[r] / PLANE_PX < p.n;
                float* yb = p.y + ((size_t)(live ? n0 + ypix[r] / PLANE_PX : n0) * p.m + mt * TM) * plane + ypix[r] % PLANE_PX;
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int m = hf * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                        const float v = TERMS == 4 ? __builtin_ldexpf(acc[r][hf][e], unscale_exponent(ex, ew)) : acc[r][hf][e];
                        if (live) {
                            if (ksplit > 1) atomicAdd(yb + (size_t)m * plane, v);
                            else yb[(size_t)m * plane] = v;
                        }
                        acc[r][hf][e] = 0.f;
                    }
            }
        }
        if (!more) break;
        __syncthreads();
        store_chunk(s);
        __syncthreads();
        tile = ntile; c = nc; mt = nmt; n0 = nn0;
        if (c == (tile % ksplit) * cps) c_end = c + cps;      // a new unit starts: its own K slice
    }
}

// Weight re-layout: fp32 -> bf16 hi/lo in [m tile][k chunk][hl][tap][octet][64 m][8 k].
// mode 0: wgt(m,k,ky,kx) = w[m][k][ky][kx]            (forward;  w is [M, K, 3, 3])
// mode 1: wgt(m,k,ky,kx) = w[k][m][2-ky][2-kx]        (data gradient of the same layer; w is [K, M, 3, 3])
// mode 2: wgt(m,k,ky,kx) = w[k][m][ky][kx]            (transposed convolution;         w is [K, M, 3, 3])
// terms = 4: fp16 hi/lo of w * 2^(14 - e), e from the bound `w_amax` (sgv_split.h); terms = 2: w rounded to fp16 (the operand format of fp16 tensors).
__global__ __launch_bounds__(256) void conv3x3_prep_weights(const float* w, u32x4* out, int m_total, int k_total, int mode, int terms, const float* w_amax = nullptr) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // one 16-B output word (8 k) of the hi plane
    const int chunks = k_total / KC;
    const int total = ((m_total + TM - 1) / TM) * chunks * 9 * 2 * TM;     // rows beyond m_total (a half-full last tile) are zero weights
    if (idx >= total) return;
    int r = idx;
    const int mi = r % TM; r /= TM;
    const int oct = r % 2; r /= 2;
    const int tap = r % 9; r /= 9;
    const int c = r % chunks;
    const int mt = r / chunks;
    const int m = mt * TM + mi, k0 = c * KC + 8 * oct, ky = tap / 3, kx = tap % 3;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
        v[j] = m >= m_total ? 0.f
             : mode == 0 ? w[(((size_t)m * k_total + k0 + j) * 3 + ky) * 3 + kx]
             : mode == 1 ? w[(((size_t)(k0 + j) * m_total + m) * 3 + (2 - ky)) * 3 + (2 - kx)]
                         : w[(((size_t)(k0 + j) * m_total + m) * 3 + ky) * 3 + kx];
    u32x4 hi, lo;
    if (terms == 4) split8t<4>(v, split_scale(amax_exponent(*w_amax)), hi, lo);
    else if (terms == 2) split8t<2>(v, 1.f, hi, lo);      // fp16 tensors: the weights rounded to fp16 (networks.py:50-52)
    else split8(v, hi, lo);
    const size_t base = ((size_t)mt * chunks + c) * WS_WORDS;
    out[base + ((0 * 9 + tap) * 2 + oct) * TM + mi] = hi;
    if (terms > 2) out[base + ((1 * 9 + tap) * 2 + oct) * TM + mi] = lo;
}


// max |v| of a tensor as an fp32 bit pattern (non-negative floats order like unsigned integers); NaN / inf propagate as the largest patterns.
// `out` must hold 0 (or a previous bound to extend) on entry.  A streaming read: every lane keeps four 16-byte loads in flight, up to 1024 workgroups
// walk the tensor grid-stride; one no-return atomic per workgroup.
// largest |element| of one dword of a tensor of T, as an fp32 bit pattern
template <typename T> __device__ __forceinline__ unsigned absmax_word(unsigned w);
template <> __device__ __forceinline__ unsigned absmax_word<float>(unsigned w) { return w & 0x7fffffffu; }
template <> __device__ __forceinline__ unsigned absmax_word<__bf16>(unsigned w) { return max((w << 16) & 0x7fffffffu, w & 0x7fff0000u); }   // a bf16 IS the upper half of the fp32 pattern
template <> __device__ __forceinline__ unsigned absmax_word<_Float16>(unsigned w) {
    const f16x2 hh = __builtin_bit_cast(f16x2, w);
    return max(__builtin_bit_cast(unsigned, (float)hh[0]) & 0x7fffffffu, __builtin_bit_cast(unsigned, (float)hh[1]) & 0x7fffffffu);
}

template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* x, size_t n, unsigned* out) {
    unsigned m = 0u;
    constexpr int EPV = 16 / sizeof(T);                       // elements per 16-byte vector
    const size_t head = min(n, (size_t)(((16 - ((uintptr_t)x & 15)) & 15) / sizeof(T)));   // elements in front of the first 16-byte boundary
    const u32x4* v = (const u32x4*)(x + head);
    const size_t nv = (n - head) / EPV;
    auto fold = [&](u32x4 w) { m = max(max(m, absmax_word<T>(w[0])), max(max(absmax_word<T>(w[1]), absmax_word<T>(w[2])), absmax_word<T>(w[3]))); };
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += 4 * stride) {
        const bool k1 = i + stride < nv, k2 = i + 2 * stride < nv, k3 = i + 3 * stride < nv;
        const u32x4 w0 = v[i], w1 = v[k1 ? i + stride : i], w2 = v[k2 ? i + 2 * stride : i], w3 = v[k3 ? i + 3 * stride : i];
        fold(w0); fold(w1); fold(w2); fold(w3);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {                 // the unaligned head and the tail shorter than a vector: at most 2 * EPV - 2 elements
        const size_t tail0 = head + nv * EPV;
        const size_t k = threadIdx.x < head ? threadIdx.x : tail0 + (threadIdx.x - head);
        if (k < n && (threadIdx.x < head || k >= tail0)) m = max(m, __builtin_bit_cast(unsigned, (float)x[k]) & 0x7fffffffu);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m) atomicMax(out, m);          // one no-return atomic per workgroup: <= 1024 per launch
    }
}

}  // namespace sgv_conv
