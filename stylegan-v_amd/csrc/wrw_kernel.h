// Weight gradient of a 3x3 / stride 1 / pad 1 convolution on NCHW fp32 tensors, on the gfx950 matrix cores:
//
//     dw[o,i,ky,kx] = sum_{n,y,x} dy[n,o,y,x] * x[n,i,y+ky-1,x+kx-1]
//
// Reference: the weight gradient of the `conv2d` calls of SynthesisLayer / Conv2dLayer (src/training/networks.py:58-62,
// layers.py:188-192 through conv2d_resample.py:40-54 and conv2d_gradfix.py:112-170 `Conv2dGradWeight`), which the reference
// hands to cuDNN.  On gfx950 MIOpen runs it as an NHWC implicit GEMM between three layout transposes (profiles/
// r01_bench_step_kernel_stats_v2.csv: 16 % + 5 % of the train step).  NCHW is the natural layout for this contraction: the
// reduction index (pixels) is the contiguous one for BOTH operands, which is exactly what an MFMA wants per lane.
//
// Arithmetic: fp32 emulated on the bf16 matrix pipe ("bf16x3"): every fp32 value is split v = hi + lo (hi = RNE bf16(v),
// lo = RNE bf16(v - hi)), and a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi with fp32 accumulation in the MFMA.  The dropped terms
// are <= 2^-16 relative per product (measured against fp64 in tests/test_conv_wrw_gpu.py beside MIOpen's own fp32 error);
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA, so three of them are still 5.3x faster and the kernel is
// bounded by LDS / HBM rather than by the matrix pipe.  TERMS = 1 gives the plain bf16 product.
//
// Work decomposition.  Output tile per workgroup: 64 o x 64 i x 9 taps (4 waves as 2x2, each 32 o x 32 i x 9 = nine 32x32
// accumulators = 144 registers).  The reduction (n, y, x) is cut into units of (one sample, one 32-pixel column segment,
// ROWS consecutive rows); a persistent grid of tiles x splits workgroups walks the units, keeps its accumulators in
// registers throughout and adds them to dw with one atomicAdd per element at the very end.
// Per row step (K = 32 pixels): dy[64 o][32 px] and ONE new x row [64 i][34 px] (32 + halo) are loaded (16-B loads, every
// 128-B line fully used), split into hi/lo bf16 and written to LDS; the 3 x rows a step needs live in a 4-slot ring, so x is
// read from HBM once (+2 halo rows per unit).  The nine taps are nine views of the same LDS rows: ky picks the ring slot,
// kx = 1 is the aligned 16-B read, kx = 0 / 2 are the same dwords shifted by one bf16 with v_alignbyte_b32.  One barrier per
// step; global loads for step y+1 are in flight during the 54 MFMAs of step y.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgv_split.h"

namespace sgv_wrw {
using sgv_conv::split2;
using sgv_conv::split8t;
using sgv_conv::mma16;
using sgv_conv::operand_exponent;
using sgv_conv::split_scale;
using sgv_conv::unscale_exponent;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TO = 64, TI = 64;   // output tile (o x i) per workgroup
constexpr int SEG = 32;           // pixels per row step
constexpr int RS = 40;            // LDS row stride in bf16 (80 B = 5 x 16 B: consecutive rows land in distinct 16-B bank groups)
constexpr int XROW0 = 8;          // LDS position of the segment's first pixel inside an x row (left halo at 7, right halo at 40)
constexpr int XS_SLOT = TI * RS + 8;

struct wrw_params {
    const float* dy;   // [n, o, h, w]
    const float* x;    // [n, i, h, w]
    float* dw;         // [o, i, 3, 3], accumulated with atomics
    int n, o, i, h, w;
    int rows;          // rows per unit
    int tiles_i;       // i / 64
    int splits;        // workgroups per output tile
    int units;         // n * (w / 32) * (h / rows)
    const float* xscale;   // [n, i] or NULL: x[n,i,:,:] is multiplied by it on its way into LDS (the styles of a modulated layer, networks.py:66; producer / consumer kernel only)
    int scatter_flush;     // 1: the element-per-lane flush (every lane of an atomic instruction in a different cache line) instead of flush_tile
    // TERMS = 4 (block-scaled fp16 split, sgv_split.h; producer / consumer kernels): bounds of max |dy|, max |x| (x_amax2: optional second factor, the bound of xscale)
    const float* dy_amax;
    const float* x_amax;
    const float* x_amax2;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));  // v_cvt_pk_bf16_f32 (RNE)
}

// End-of-kernel flush of one wave's nine 32 x 32 accumulators (row = first weight index, column = second) into dw[rows][ld][9] with atomics.
// The MFMA layout puts the 64 lanes of an accumulator register into 64 different cache lines of dw (lane -> column: 36 B apart, lane half -> row);
// 256 workgroups x 36,864 such atomics at the same moment cost 0.15-0.18 ms per launch (tools/wrw_lab.hip, WRW_ABL=8: 12 % of the kernel).  So the tile
// goes through LDS, eight weight rows (8 x 32 columns x 9 taps = 2,304 consecutive-per-row floats) at a time, and every atomic instruction adds 64
// CONSECUTIVE floats.  `stage` = 2,304 floats of LDS private to this wave: the only ordering needed is the wave's own (LDS operations of a wave are in order).
constexpr int FLUSH_STAGE_FLOATS = 8 * 32 * 9;
__device__ __forceinline__ void flush_tile(const f32x16 (&acc)[9], float* stage, float* dw, int ld, int row0, int col0) {
    const int lane = threadIdx.x & 63, r32 = lane & 31, g = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; q++) {     // C layout of the 32x32 MFMA: row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): registers 4q .. 4q+3 hold rows 8q .. 8q+7
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the reads of the previous eight rows have their data
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 9; k++)
#pragma unroll
            for (int e = 0; e < 4; e++) stage[(e + 4 * g) * 288 + r32 * 9 + k] = acc[k][4 * q + e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 6
        for (int t = 0; t < 36; t++) {
            const int f = t * 64 + lane, r = f / 288, c = f - r * 288;
            atomicAdd(dw + ((size_t)(row0 + 8 * q + r) * ld + col0) * 9 + c, stage[f]);
        }
    }
}

// 8 fp32 -> 8 bf16 hi (4 dwords) + 8 bf16 lo.
__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned h = pack_bf16(v[2 * k], v[2 * k + 1]);
        const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
        hi[k] = h;
        lo[k] = pack_bf16(v[2 * k] - h0, v[2 * k + 1] - h1);
    }
}

struct row_regs {
    float main[8];   // 8 consecutive pixels of row (t / 4), starting at 8 * (t % 4)
    float halo;      // threads < 128: column x0 - 1 (even t) or x0 + 32 (odd t) of row t / 2
};

template <int TERMS>
__global__ __launch_bounds__(256, 2) void wrw3x3_kernel(wrw_params p) {
    // [hi/lo][ring slot][i][RS] and [hi/lo][buffer][o][RS]
    __shared__ __attribute__((aligned(16))) unsigned short xs[2][4][XS_SLOT];
    __shared__ __attribute__((aligned(16))) unsigned short ds[2][2][TO * RS];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wo = (wave >> 1) * 32, wi = (wave & 1) * 32;
    const int r32 = lane & 31, g = lane >> 5;

    // XCD-aware: workgroups that walk the same units (same split) on different output tiles share one L2
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int vid = (nwg & 7) == 0 ? (lin & 7) * (nwg >> 3) + (lin >> 3) : lin;
    const int tile = vid % (int)gridDim.x, split = vid / (int)gridDim.x;
    const int o0 = (tile / p.tiles_i) * TO, i0 = (tile % p.tiles_i) * TI;
    const int segs = p.w / SEG, rblocks = p.h / p.rows;
    const size_t plane = (size_t)p.h * p.w;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[k][e] = 0.f;

    const int lr = t >> 2, lq = (t & 3) * 8;   // loader role: row (channel) and first pixel
    const int hr = t >> 1, hside = t & 1;      // halo loader role (t < 128)

    for (int u = split; u < p.units; u += p.splits) {
        const int rb = u % rblocks, sg = (u / rblocks) % segs, n = u / (rblocks * segs);
        const int y0 = rb * p.rows, x0 = sg * SEG;
        const float* dyb = p.dy + ((size_t)n * p.o + o0) * plane + x0;
        const float* xb = p.x + ((size_t)n * p.i + i0) * plane + x0;

        auto load_x = [&](int row, row_regs& r) {
            if (row >= 0 && row < p.h) {
                const float* q = xb + (size_t)lr * plane + (size_t)row * p.w + lq;
                const f32x4 a = *(const f32x4*)q, b = *(const f32x4*)(q + 4);
#pragma unroll
                for (int k = 0; k < 4; k++) { r.main[k] = a[k]; r.main[4 + k] = b[k]; }
                r.halo = 0.f;
                if (t < 128) {
                    const int col = hside ? x0 + SEG : x0 - 1;
                    if (col >= 0 && col < p.w) r.halo = xb[(size_t)hr * plane + (size_t)row * p.w + (col - x0)];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) r.main[k] = 0.f;
                r.halo = 0.f;
            }
        };
        auto store_x = [&](int row, const row_regs& r) {
            const int slot = (row + 1) & 3;
            u32x4 hi, lo;
            split8(r.main, hi, lo);
            *(u32x4*)&xs[0][slot][lr * RS + XROW0 + lq] = hi;
            *(u32x4*)&xs[1][slot][lr * RS + XROW0 + lq] = lo;
            if (t < 128) {
                const unsigned h = pack_bf16(r.halo, 0.f);
                const float hf = __builtin_bit_cast(float, h << 16);
                const unsigned l = pack_bf16(r.halo - hf, 0.f);
                const int pos = hr * RS + (hside ? XROW0 + SEG : XROW0 - 1);
                xs[0][slot][pos] = (unsigned short)h;
                xs[1][slot][pos] = (unsigned short)l;
            }
        };
        auto load_dy = [&](int row, float* v) {
            const float* q = dyb + (size_t)lr * plane + (size_t)row * p.w + lq;
            const f32x4 a = *(const f32x4*)q, b = *(const f32x4*)(q + 4);
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = a[k]; v[4 + k] = b[k]; }
        };
        auto store_dy = [&](int row, const float* v) {
            u32x4 hi, lo;
            split8(v, hi, lo);
            *(u32x4*)&ds[0][row & 1][lr * RS + lq] = hi;
            *(u32x4*)&ds[1][row & 1][lr * RS + lq] = lo;
        };

        // Prologue: rows y0-1, y0, y0+1 of x and row y0 of dy (all loads in flight together).
        {
            row_regs ra, rb_, rc;
            float dv[8];
            load_x(y0 - 1, ra);
            load_x(y0, rb_);
            load_x(y0 + 1, rc);
            load_dy(y0, dv);
            __syncthreads();   // the previous unit's last step has been consumed by every wave
            store_x(y0 - 1, ra);
            store_x(y0, rb_);
            store_x(y0 + 1, rc);
            store_dy(y0, dv);
            __syncthreads();
        }

        for (int y = y0; y < y0 + p.rows; y++) {
            const bool more = y + 1 < y0 + p.rows;
            row_regs rn;
            float dn[8];
            if (more) {   // lands during the MFMAs below
                load_x(y + 2, rn);
                load_dy(y + 1, dn);
            }

            const int buf = y & 1;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int pa = (wo + r32) * RS + 16 * c + 8 * g;
                const u32x4 a_hi = *(const u32x4*)&ds[0][buf][pa];
                u32x4 a_lo;
                if (TERMS > 1) a_lo = *(const u32x4*)&ds[1][buf][pa];
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    const int slot = (y + ky) & 3;   // ring slot of row y + ky - 1
                    const int pb = (wi + r32) * RS + XROW0 + 16 * c + 8 * g;
                    u32x4 b[2][3];   // [hi/lo][kx]
#pragma unroll
                    for (int hl = 0; hl < (TERMS > 1 ? 2 : 1); hl++) {
                        const u32x4 d = *(const u32x4*)&xs[hl][slot][pb];
                        const unsigned dm = *(const unsigned*)&xs[hl][slot][pb - 2];
                        const unsigned da = *(const unsigned*)&xs[hl][slot][pb + 8];
                        const unsigned a01 = __builtin_amdgcn_alignbyte(d[1], d[0], 2), a12 = __builtin_amdgcn_alignbyte(d[2], d[1], 2),
                                       a23 = __builtin_amdgcn_alignbyte(d[3], d[2], 2);
                        b[hl][0] = u32x4{__builtin_amdgcn_alignbyte(d[0], dm, 2), a01, a12, a23};
                        b[hl][1] = d;
                        b[hl][2] = u32x4{a01, a12, a23, __builtin_amdgcn_alignbyte(da, d[3], 2)};
                    }
                    // small terms first; consecutive MFMAs target different accumulators
                    if (TERMS > 1) {
#pragma unroll
                        for (int kx = 0; kx < 3; kx++)
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_lo), __builtin_bit_cast(bf16x8, b[0][kx]),
                                                                                       acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
                        for (int kx = 0; kx < 3; kx++)
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hi), __builtin_bit_cast(bf16x8, b[1][kx]),
                                                                                       acc[ky * 3 + kx], 0, 0, 0);
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; kx++)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hi), __builtin_bit_cast(bf16x8, b[0][kx]),
                                                                                   acc[ky * 3 + kx], 0, 0, 0);
                }
            }

            if (more) {
                store_x(y + 2, rn);     // slot of row y-2: its last readers (step y-1) are behind the previous barrier
                store_dy(y + 1, dn);
            }
            __syncthreads();
        }
    }

    // Flush (every wave is behind the loop's last barrier: the LDS tiles are free).
    if (!p.scatter_flush) { flush_tile(acc, (float*)&xs[0][0][0] + (threadIdx.x >> 6) * FLUSH_STAGE_FLOATS, p.dw, p.i, o0 + wo, i0 + wi); return; }
    // C layout of the 32x32 MFMA: col (i) = lane & 31, row (o) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int o = o0 + wo + (e & 3) + 8 * (e >> 2) + 4 * g;
            const int i = i0 + wi + r32;
            atomicAdd(p.dw + ((size_t)o * p.i + i) * 9 + k, acc[k][e]);
        }
}

// ----------------------------------------------------------------------------------------------------------------------
// Stride-2 member: weight gradient of the strided (and, by symmetry, of the transposed) 3x3 convolution that connects a
// "big" tensor [n, cb, 2H+1, 2W+1] with a "small" one [n, cs, H, W] (conv3x3s2_kernel.h):
//
//     dw[cs, cb, ky, kx] = sum_{n,Y,X} small[n,cs,Y,X] * big[n,cb,2Y+ky,2X+kx]
//
// (strided layer: small = dy, big = x, dw has the weight's [c_out, c_in, 3, 3] layout; transposed layer: small = x, big = dy,
// dw has the weight's [c_in, c_out, 3, 3] layout -- the same formula.)  Same structure as wrw3x3_kernel; differences: a step
// consumes two new big rows (ring of 5 row slots), big rows have no 16-B alignment (4-B aligned dwordx4 loads), and the
// columns are de-interleaved into even / odd planes on the way into LDS so that kx = 0 / 1 are aligned 16-B reads of the
// even / odd plane and kx = 2 is the even plane shifted by one bf16 (v_alignbyte_b32).
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BIG_CH = 2 * RS + 8;           // bf16 per channel row in a ring slot: even plane [RS] + odd plane [RS] + pad (176 B = 11 x 16 B)
constexpr int BIG_SLOT = TI * BIG_CH;        // bf16 per ring slot

struct wrw_s2_params {
    const float* small;   // [n, cs, h, w]
    const float* big;     // [n, cb, 2h+1, 2w+1]
    float* dw;            // [cs, cb, 3, 3]
    int n, cs, cb, h, w;
    int rows, tiles_b, splits, units;
    int scatter_flush;    // as in wrw_params
    const float* small_amax;   // TERMS = 4 (block-scaled fp16 split, sgv_split.h; producer / consumer kernel): bounds of max |small|, max |big|
    const float* big_amax;
};

//
// PACK: small images 16 or 8 pixels wide (big: 33 / 17): 2 or 4 samples side by side in the 32-pixel row step, a unit is (group of 32 / W samples, row
// block).  Every sample's last big column (2W, the kx = 2 neighbour of its last pixel) sits in the pad words of the channel row (position 80 + 2 * sample)
// instead of behind the even plane, and the lane groups that end a sample take it from there.
template <int TERMS, bool PACK = false>
__global__ __launch_bounds__(256, 1) void wrw3x3_s2_kernel(wrw_s2_params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_s2[];
    unsigned short* bs = lds_s2;                          // [hl][5 slots][64 cb][even RS | odd RS]
    unsigned short* as = lds_s2 + 2 * 5 * BIG_SLOT;       // [hl][2 bufs][64 cs][RS]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wo = (wave >> 1) * 32, wi = (wave & 1) * 32;
    const int r32 = lane & 31, g = lane >> 5;
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int vid = (nwg & 7) == 0 ? (lin & 7) * (nwg >> 3) + (lin >> 3) : lin;
    const int tile = vid % (int)gridDim.x, split = vid / (int)gridDim.x;
    const int s0 = (tile / p.tiles_b) * TO, b0 = (tile % p.tiles_b) * TI;
    const int segs = PACK ? 1 : p.w / SEG, rblocks = p.h / p.rows;
    const int wsh = PACK ? 31 - __builtin_clz(p.w) : 5;   // log2(W): W is 16 or 8 when PACK
    const int spr = PACK ? SEG >> wsh : 1;     // samples per row step
    // offset (inside a channel row of a ring slot) of the dword behind this lane's eight even columns, per k half c; PACK: where the group ends a
    // sample, that sample's last big column (kept in the pad words)
    int pe_off[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int px = 16 * c + 8 * (lane >> 5);
        pe_off[c] = px + 8;
        if (PACK && ((px + 8) & (p.w - 1)) == 0) pe_off[c] = 2 * RS + 2 * (px >> wsh);
    }
    const int hb = 2 * p.h + 1, wb = 2 * p.w + 1;
    const size_t plane_s = (size_t)p.h * p.w, plane_b = (size_t)hb * wb;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[k][e] = 0.f;

    const int lr = t >> 2, lq = (t & 3) * 8;   // small-row loader: channel, first pixel

    struct big_regs { f32x4 v[4]; float edge; };

    for (int u = split; u < p.units; u += p.splits) {
        const int rb = u % rblocks, sg = (u / rblocks) % segs, n = (u / (rblocks * segs)) * spr;
        const int y0 = rb * p.rows, x0 = sg * SEG;
        const float* sb = p.small + ((size_t)n * p.cs + s0) * plane_s + x0;
        const float* bb = p.big + ((size_t)n * p.cb + b0) * plane_b + (size_t)(2 * y0) * wb + 2 * x0;
        // PACK: sample `s` of the group (clamped to the batch; a sample past its end contributes zeros through the small operand)
        auto sample_off = [&](int smp, size_t per_sample) { return (size_t)min(smp, p.n - 1 - n) * per_sample; };

        auto load_big = [&](int b, big_regs& r) {   // local big row b = 0 .. 2 * rows
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int it = t + 256 * j, quad = it & 15, ch = it >> 4;
                if (PACK) {   // quads per sample: W / 2
                    const int smp = quad >> (wsh - 1), col = 4 * (quad & ((p.w >> 1) - 1));
                    r.v[j] = *(const f32x4_u*)(bb + sample_off(smp, (size_t)p.cb * plane_b) + (size_t)ch * plane_b + (size_t)b * wb + col);
                } else r.v[j] = *(const f32x4_u*)(bb + (size_t)ch * plane_b + (size_t)b * wb + 4 * quad);
            }
            if (PACK) r.edge = t < TI * spr ? bb[sample_off(t >> 6, (size_t)p.cb * plane_b) + (size_t)(t & 63) * plane_b + (size_t)b * wb + 2 * p.w] : 0.f;
            else r.edge = t < TI ? bb[(size_t)t * plane_b + (size_t)b * wb + 64] : 0.f;
        };
        auto store_big = [&](int b, const big_regs& r) {
            const int slot = b - 5 * ((b * 205) >> 10);   // b % 5
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int it = t + 256 * j, quad = it & 15, ch = it >> 4;
                const unsigned he = pack_bf16(r.v[j][0], r.v[j][2]), ho = pack_bf16(r.v[j][1], r.v[j][3]);
                const int pos = slot * BIG_SLOT + ch * BIG_CH + 2 * quad;
                *(unsigned*)&bs[pos] = he;
                *(unsigned*)&bs[pos + RS] = ho;
                if (TERMS > 1) {
                    const unsigned le = pack_bf16(r.v[j][0] - __builtin_bit_cast(float, he << 16), r.v[j][2] - __builtin_bit_cast(float, he & 0xffff0000u));
                    const unsigned lo = pack_bf16(r.v[j][1] - __builtin_bit_cast(float, ho << 16), r.v[j][3] - __builtin_bit_cast(float, ho & 0xffff0000u));
                    *(unsigned*)&bs[5 * BIG_SLOT + pos] = le;
                    *(unsigned*)&bs[5 * BIG_SLOT + pos + RS] = lo;
                }
            }
            if (t < TI * spr) {
                const unsigned h = pack_bf16(r.edge, 0.f);
                const int pos = slot * BIG_SLOT + (t & 63) * BIG_CH + (PACK ? 2 * RS + 2 * (t >> 6) : 32);
                bs[pos] = (unsigned short)h;
                if (TERMS > 1) bs[5 * BIG_SLOT + pos] = (unsigned short)pack_bf16(r.edge - __builtin_bit_cast(float, h << 16), 0.f);
            }
        };
        auto load_small = [&](int row, float* v) {
            const int smp = PACK ? lq >> wsh : 0;
            const float* q = sb + (size_t)lr * plane_s + (size_t)(y0 + row) * p.w + (PACK ? (lq & (p.w - 1)) : lq);
            if (PACK) q += sample_off(smp, (size_t)p.cs * plane_s);
            const f32x4 a = *(const f32x4*)q, b = *(const f32x4*)(q + 4);
            const bool live = !PACK || n + smp < p.n;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = live ? a[k] : 0.f; v[4 + k] = live ? b[k] : 0.f; }
        };
        auto store_small = [&](int row, const float* v) {
            u32x4 hi, lo;
            split8(v, hi, lo);
            *(u32x4*)&as[(row & 1) * (TO * RS) + lr * RS + lq] = hi;
            if (TERMS > 1) *(u32x4*)&as[2 * TO * RS + (row & 1) * (TO * RS) + lr * RS + lq] = lo;
        };

        {
            big_regs r0, r1, r2;
            float sv[8];
            load_big(0, r0);
            load_big(1, r1);
            load_big(2, r2);
            load_small(0, sv);
            __syncthreads();
            store_big(0, r0);
            store_big(1, r1);
            store_big(2, r2);
            store_small(0, sv);
            __syncthreads();
        }

        for (int i = 0; i < p.rows; i++) {
            const bool more = i + 1 < p.rows;
            big_regs rn0, rn1;
            float sn[8];
            if (more) {
                load_big(2 * i + 3, rn0);
                load_big(2 * i + 4, rn1);
                load_small(i + 1, sn);
            }
            const int buf = i & 1;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int pa = buf * (TO * RS) + (wo + r32) * RS + 16 * c + 8 * g;
                const u32x4 a_hi = *(const u32x4*)&as[pa];
                u32x4 a_lo;
                if (TERMS > 1) a_lo = *(const u32x4*)&as[2 * TO * RS + pa];
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    const int b = 2 * i + ky;
                    const int slot = b - 5 * ((b * 205) >> 10);
                    const int pb = slot * BIG_SLOT + (wi + r32) * BIG_CH + 16 * c + 8 * g;
                    const int pe = slot * BIG_SLOT + (wi + r32) * BIG_CH + pe_off[c];
                    u32x4 bv[2][3];
#pragma unroll
                    for (int hl = 0; hl < (TERMS > 1 ? 2 : 1); hl++) {
                        const u32x4 e = *(const u32x4*)&bs[hl * 5 * BIG_SLOT + pb];
                        const unsigned ea = *(const unsigned*)&bs[hl * 5 * BIG_SLOT + pe];
                        bv[hl][0] = e;
                        bv[hl][1] = *(const u32x4*)&bs[hl * 5 * BIG_SLOT + pb + RS];
                        bv[hl][2] = u32x4{__builtin_amdgcn_alignbyte(e[1], e[0], 2), __builtin_amdgcn_alignbyte(e[2], e[1], 2), __builtin_amdgcn_alignbyte(e[3], e[2], 2),
                                          __builtin_amdgcn_alignbyte(ea, e[3], 2)};
                    }
                    if (TERMS > 1) {
#pragma unroll
                        for (int kx = 0; kx < 3; kx++)
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_lo), __builtin_bit_cast(bf16x8, bv[0][kx]), acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
                        for (int kx = 0; kx < 3; kx++)
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hi), __builtin_bit_cast(bf16x8, bv[1][kx]), acc[ky * 3 + kx], 0, 0, 0);
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; kx++)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hi), __builtin_bit_cast(bf16x8, bv[0][kx]), acc[ky * 3 + kx], 0, 0, 0);
                }
            }
            if (more) {
                store_big(2 * i + 3, rn0);
                store_big(2 * i + 4, rn1);
                store_small(i + 1, sn);
            }
            __syncthreads();
        }
    }

    if (!p.scatter_flush) { flush_tile(acc, (float*)lds_s2 + (threadIdx.x >> 6) * FLUSH_STAGE_FLOATS, p.dw, p.cb, s0 + wo, b0 + wi); return; }
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int cs = s0 + wo + (e & 3) + 8 * (e >> 2) + 4 * g;
            const int cb = b0 + wi + r32;
            atomicAdd(p.dw + ((size_t)cs * p.cb + cb) * 9 + k, acc[k][e]);
        }
}

constexpr int WRW_S2_LDS_BYTES = (2 * 5 * BIG_SLOT + 2 * 2 * TO * RS) * 2;

}  // namespace sgv_wrw
