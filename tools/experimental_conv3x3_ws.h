// EXPERIMENTAL -- not part of libsgv_hip.so, never run on hardware yet (written after the round-1 GPU budget was spent).
// Producer / consumer ("warp-specialised") variant of conv3x3_kernel (stylegan-v_amd/csrc/conv3x3_kernel.h), to be
// validated and timed with tools/conv_lab.hip (section "ws") before it replaces anything.
//
// Motivation (profiles/r01_conv3x3_ablation.log, 128ch 128^2, 1.47 ms): compute-only 0.95 ms, load + split + LDS fill
// alone 0.77 ms, and the two phases are serialised by the barrier pair of the 4-wave kernel.  Here a workgroup has 8 waves:
// waves 0-3 only issue LDS operand reads + MFMAs (128 accumulators each, as before), waves 4-7 only load the next chunk,
// split it into hi/lo bf16 and write it into the OTHER of two LDS images.  Every SIMD then hosts one MFMA wave and one
// producer wave, whose VALU / LDS-write work overlaps the MFMAs.  One barrier per 16-channel chunk:
//
//     iteration q:   consumers: MFMAs on image q & 1          producers: store chunk q+1 -> image (q+1) & 1, load chunk q+2
//     barrier        (image (q+1)&1 was last read in iteration q-1, i.e. before the previous barrier)
//
// LDS: 2 x (x tile 38.3 KiB + weights 36 KiB) = 148.5 KiB (limit 160 KiB) -> one workgroup per CU, 512 threads, <= 256 registers.
// Known open point: the compiler allocates for the union of both roles and spills ~39 registers at the 256 limit (moving the
// weight staging to the consumer waves made it worse, 150); if the variant pays off, give the roles their own budgets
// (accumulators pinned to AGPRs, or weights by global->LDS DMA so that the producers need no weight registers at all).
#pragma once

#include "conv3x3_kernel.h"

namespace sgv_conv {

constexpr int WS_IMAGE_WORDS = XS_WORDS + WS_WORDS;
constexpr int WS_LDS_BYTES = 2 * WS_IMAGE_WORDS * 16;

template <int TERMS>
__global__ __launch_bounds__(512, 1) void conv3x3_ws_kernel(conv_params p) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool producer = wave >= 4;
    const int pt = t & 255;                      // thread index inside its role group
    const int cw = wave & 3;                     // consumer wave index (rows 4 * cw ..)
    const int l32 = lane & 31, g = lane >> 5;
    const int chunks = p.k / KC;
    const size_t plane = (size_t)p.h * p.w;
    constexpr int ITEMS = 16 * RIN;

    // producer roles (same mapping as conv3x3_kernel, indexed by pt)
    const int a_oct = pt & 1, a_quad = (pt >> 1) & 7, a_row = pt >> 4;
    const int b_row = 16 + (pt >> 4);
    const int h_oct = pt & 1, h_side = (pt >> 1) & 1, h_row = pt >> 2;

    auto load_chunk = [&](const tile_pos& tp, int c, stage_regs& s) {
        const float* xb = p.x + ((size_t)tp.n * p.k + c * KC) * plane + tp.x0;
        {
            const int gy = tp.y0 - 1 + a_row;
            const bool ok = gy >= 0 && gy < p.h;
            const float* q = xb + (size_t)(8 * a_oct) * plane + (size_t)gy * p.w + 4 * a_quad;
#pragma unroll
            for (int j = 0; j < 8; j++) s.xa[j] = ok ? *(const f32x4*)(q + j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (pt + 256 < ITEMS) {
            const int gy = tp.y0 - 1 + b_row;
            const bool ok = gy < p.h;
            const float* q = xb + (size_t)(8 * a_oct) * plane + (size_t)gy * p.w + 4 * a_quad;
#pragma unroll
            for (int j = 0; j < 8; j++) s.xb[j] = ok ? *(const f32x4*)(q + j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (pt < 4 * RIN) {
            const int gy = tp.y0 - 1 + h_row;
            const int gx = h_side ? tp.x0 + SEG : tp.x0 - 1;
            const bool ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
            const float* q = xb + (size_t)(8 * h_oct) * plane + (size_t)gy * p.w + (gx - tp.x0);
#pragma unroll
            for (int j = 0; j < 8; j++) s.xh[j] = ok ? q[j * plane] : 0.f;
        }
        const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + c) * WS_WORDS + pt;
#pragma unroll
        for (int j = 0; j < 9; j++) s.wv[j] = wq[j * 256];
    };
    auto store_chunk = [&](u32x4* xs, u32x4* ws, const stage_regs& s) {
        {
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
        if (pt + 256 < ITEMS) {
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
        if (pt < 4 * RIN) {
            u32x4 hi, lo;
            split8(s.xh, hi, lo);
            const int pos = (h_oct * RIN + h_row) * PIN + (h_side ? SEG + 1 : 0);
            xs[pos] = hi;
            if (TERMS > 1) xs[2 * XS_PLANE + pos] = lo;
        }
#pragma unroll
        for (int j = 0; j < 9; j++) ws[pt + j * 256] = s.wv[j];
    };

    // flat sequence of (tile, chunk) pairs this workgroup walks: q -> tile = first + (q / chunks) * grid, chunk = q % chunks
    const int first = xcd_swizzle(blockIdx.x, gridDim.x);
    if (first >= p.tiles) return;
    const int my_tiles = (p.tiles - first + p.grid - 1) / p.grid;
    const int total = my_tiles * chunks;

    if (producer) {
        // chunk 0 -> image 0 (synchronously), chunk 1 in registers
        stage_regs s;
        tile_pos tp = decode_tile(p, first, TROWS);
        load_chunk(tp, 0, s);
        store_chunk(lds, lds + XS_WORDS, s);
        if (total > 1) {
            const int t1 = first + (1 / chunks) * p.grid;
            tile_pos tp1 = decode_tile(p, t1, TROWS);
            load_chunk(tp1, 1 % chunks, s);
        }
        __syncthreads();   // image 0 ready
        for (int q = 0; q < total; q++) {
            if (q + 1 < total) {
                u32x4* img = lds + ((q + 1) & 1) * WS_IMAGE_WORDS;
                store_chunk(img, img + XS_WORDS, s);   // chunk q+1 (loaded during iteration q-1)
                if (q + 2 < total) {
                    const int tq = first + ((q + 2) / chunks) * p.grid;
                    tile_pos tpq = decode_tile(p, tq, TROWS);
                    load_chunk(tpq, (q + 2) % chunks, s);
                }
            }
            __syncthreads();
        }
        return;
    }

    // ---- consumers ----
    f32x16 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][hf][e] = 0.f;
    __syncthreads();   // image 0 ready
    for (int q = 0; q < total; q++) {
        const u32x4* xs = lds + (q & 1) * WS_IMAGE_WORDS;
        const u32x4* ws = xs + XS_WORDS;
        const int c = q % chunks;

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
            for (int rh = 0; rh < 2; rh++) {   // two rows at a time: 16 operand registers instead of 32 (the kernel has to fit 256)
                u32x4 b_hi[2], b_lo[2];
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int pos = (g * RIN + 4 * cw + 2 * rh + r + ky) * PIN + l32 + kx;
                    b_hi[r] = xs[pos];
                    if (TERMS > 1) b_lo[r] = xs[2 * XS_PLANE + pos];
                }
                if (TERMS > 1) {
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int hf = 0; hf < 2; hf++)
                            acc[2 * rh + r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][1]), __builtin_bit_cast(bf16x8, b_hi[r]), acc[2 * rh + r][hf], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int hf = 0; hf < 2; hf++)
                            acc[2 * rh + r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][0]), __builtin_bit_cast(bf16x8, b_lo[r]), acc[2 * rh + r][hf], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[2 * rh + r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[hf][0]), __builtin_bit_cast(bf16x8, b_hi[r]), acc[2 * rh + r][hf], 0, 0, 0);
            }
        }
        if (c == chunks - 1) {
            const int tile = first + (q / chunks) * p.grid;
            const tile_pos tp = decode_tile(p, tile, TROWS);
            float* yb = p.y + ((size_t)tp.n * p.m + tp.mt * TM) * plane + (size_t)(tp.y0 + 4 * cw) * p.w + tp.x0 + l32;
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int m = hf * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                        yb[(size_t)m * plane + (size_t)r * p.w] = acc[r][hf][e];
                        acc[r][hf][e] = 0.f;
                    }
        }
        __syncthreads();
    }
}

}  // namespace sgv_conv
