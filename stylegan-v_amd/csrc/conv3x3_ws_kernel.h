// 3x3 / stride 1 / pad 1 convolution on NCHW fp32 tensors -- producer / consumer form of conv3x3_kernel (conv3x3_kernel.h), same
// arithmetic (bf16 hi/lo split products on v_mfma_f32_32x32x16_bf16, fp32 accumulate), same LDS images, same weight layout.
//
// Reference: the `conv2d` of SynthesisLayer / Conv2dLayer (src/training/networks.py:58-62 via conv2d_resample.py:40-54), its data
// gradient (conv2d_gradfix.py:100-118), and -- with PRO / EPI -- the modulation multiplies and the bias_act that surround it in
// `modulated_conv2d` + SynthesisLayer.forward (networks.py:65-74, 141-143).
//
// Why a second form: in the 4-wave kernel one wave per SIMD does everything; per 16-channel chunk it issues 216 MFMAs, then all
// waves meet at a barrier, split / write the next chunk into LDS (matrix pipe idle), and meet again.  The ISA shows the other half
// of the loss: 128 accumulators + 108 staging registers + operands leave no room to prefetch MFMA operands, so the wave waits
// for LDS reads between MFMAs (profiles/r01_conv3x3_ablation.log: 9.9k cycles per chunk with NO fill at all, against 6.9k of MFMA
// issue).  Here a workgroup has 8 waves, two per SIMD:
//
//   waves 0-3 (consumers): only LDS operand reads + MFMAs on image q & 1 (128 accumulators, operands double-buffered in
//                          registers one half-tap ahead), the epilogue when a tile is complete;
//   waves 4-7 (producers): write chunk q+1 (already in registers) into image (q+1) & 1 -- optional per-(sample, channel) scale, hi/lo
//                          split, transposing ds_write_b128 -- start the weight block of chunk q+1 as global -> LDS DMA (36 x 1 KiB
//                          `global_load_lds_dwordx4`, no registers, no ds_write), load x of chunk q+2, wait, barrier.
//
// One barrier per chunk.  The producers' VALU / LDS-write / VMEM instructions issue in the gaps between the consumer's MFMAs of the
// same SIMD (separate pipes).  LDS: 2 x (x tile 38.3 KiB + weights 36 KiB + 512 B epilogue vectors) = 149.5 KiB, one workgroup per CU.
#pragma once

#include "conv3x3_kernel.h"

namespace sgv_conv {

constexpr int WS_EP_WORDS = 64;                                      // 256 floats per image: the tile's four epilogue vectors c0..c3[64]
constexpr int WS_IMAGE_WORDS = XS_WORDS + WS_WORDS + WS_EP_WORDS;    // u32x4 words per LDS image
constexpr int WS_LDS_BYTES = 2 * WS_IMAGE_WORDS * 16;

struct conv_ws_params {
    conv_params c;
    const float* xscale;   // PRO = 1: [n, k]   x[n,k,:,:] is multiplied by it before the split (styles, networks.py:66)
    const float* oscale;   // EPI >= 1: [n, m] or NULL (demodulation coefficients, networks.py:70-71)
    const float* bias;     // EPI >= 1: [m] or NULL
    int act;               // 1 linear, 3 lrelu  (bias_act.cu activation indices)
    float alpha, gain, clamp;   // clamp < 0: none
};

// PRO: 0 plain x, 1 x * xscale[n,k].
// EPI: 0 plain store; 1: y = clamp(lrelu_alpha(acc * oscale[n,m] + bias[m]) * gain), evaluated as max(fma(acc, c0, c1), fma(acc, c2, c3)) with
//      c0 = oscale * gain, c1 = bias * gain, c2 = c0 * alpha, c3 = c1 * alpha prepared per tile by the producers (valid for gain > 0,
//      0 <= alpha <= 1; alpha = 1 is the linear activation): 3 VALU operations per output in the consumer, which is what the MFMA waves
//      can afford.  Differs from the three-pass composition (networks.py:70-71 + bias_act.cu:39-146) by fused-multiply-add rounding only.
template <int TERMS, int PRO, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(conv_ws_params pp) {
    const conv_params& p = pp.c;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int chunks = p.k / KC;
    const size_t plane = (size_t)p.h * p.w;

    // flat sequence of (tile, chunk) pairs of this workgroup: q -> tile = first + (q / chunks) * grid, chunk = q % chunks
    const int first = xcd_swizzle(blockIdx.x, gridDim.x);
    if (first >= p.tiles) return;
    const int my_tiles = (p.tiles - first + p.grid - 1) / p.grid;
    const int total = my_tiles * chunks;

    if (wave >= 4) {
        // =========================================== producers ===========================================
        const int pt = t - 256, pw = wave - 4;
        constexpr int ITEMS = 16 * RIN;
        const int a_oct = pt & 1, a_quad = (pt >> 1) & 7, a_row = pt >> 4;
        const int b_row = 16 + (pt >> 4);
        const int h_oct = pt & 1, h_side = (pt >> 1) & 1, h_row = pt >> 2;
        f32x4 xa[8], xb[8];
        float xh[8];
        f32x4 sc[2];   // PRO: the 8 channel scales of this thread's octet

        auto load_x = [&](int q) {
            const tile_pos tp = decode_tile(p, first + (q / chunks) * p.grid, TROWS);
            const int c = q % chunks;
            const float* xb_ = p.x + ((size_t)tp.n * p.k + c * KC) * plane + tp.x0;
            {
                const int gy = tp.y0 - 1 + a_row;
                const bool ok = gy >= 0 && gy < p.h;
                const float* qx = xb_ + (size_t)(8 * a_oct) * plane + (size_t)gy * p.w + 4 * a_quad;
#pragma unroll
                for (int j = 0; j < 8; j++) xa[j] = ok ? *(const f32x4*)(qx + j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (pt + 256 < ITEMS) {
                const int gy = tp.y0 - 1 + b_row;
                const bool ok = gy < p.h;
                const float* qx = xb_ + (size_t)(8 * a_oct) * plane + (size_t)gy * p.w + 4 * a_quad;
#pragma unroll
                for (int j = 0; j < 8; j++) xb[j] = ok ? *(const f32x4*)(qx + j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (pt < 4 * RIN) {
                const int gy = tp.y0 - 1 + h_row;
                const int gx = h_side ? tp.x0 + SEG : tp.x0 - 1;
                const bool ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
                const float* qx = xb_ + (size_t)(8 * h_oct) * plane + (size_t)gy * p.w + (gx - tp.x0);
#pragma unroll
                for (int j = 0; j < 8; j++) xh[j] = ok ? qx[j * plane] : 0.f;
            }
            if (PRO == 1) {   // a_oct == h_oct: one octet of scales serves all three items
                const float* sp = pp.xscale + (size_t)tp.n * p.k + c * KC + 8 * a_oct;
                sc[0] = *(const f32x4*)sp;
                sc[1] = *(const f32x4*)(sp + 4);
            }
        };
        auto put = [&](u32x4* xs, int pos, float* v) {
            if (PRO == 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] *= sc[j >> 2][j & 3];
            }
            u32x4 hi, lo;
            split8(v, hi, lo);
            xs[pos] = hi;
            if (TERMS > 1) xs[2 * XS_PLANE + pos] = lo;
        };
        auto store_x = [&](u32x4* xs) {
            {
                const int base = (a_oct * RIN + a_row) * PIN + 1 + 4 * a_quad;
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = xa[j][px];
                    put(xs, base + px, v);
                }
            }
            if (pt + 256 < ITEMS) {
                const int base = (a_oct * RIN + b_row) * PIN + 1 + 4 * a_quad;
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = xb[j][px];
                    put(xs, base + px, v);
                }
            }
            if (pt < 4 * RIN) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = xh[j];
                put(xs, (h_oct * RIN + h_row) * PIN + (h_side ? SEG + 1 : 0), v);
            }
        };
        // weights of (tile, chunk) q: 36 KiB copied verbatim; each wave-instruction moves 64 lanes x 16 B to a wave-uniform LDS base
        auto dma_w = [&](int q, u32x4* img) {
            const tile_pos tp = decode_tile(p, first + (q / chunks) * p.grid, TROWS);
            const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + (q % chunks)) * WS_WORDS + pw * 64 + lane;
            u32x4* wl = img + XS_WORDS + pw * 64;
#pragma unroll
            for (int j = 0; j < 9; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wq + j * 256),
                                                 (__attribute__((address_space(3))) void*)(wl + j * 256), 16, 0, 0);
        };
        auto put_ep = [&](int q, u32x4* img) {   // the tile's epilogue vectors ride with its LAST chunk
            if (EPI == 0 || (q % chunks) != chunks - 1 || pt >= TM) return;
            const tile_pos tp = decode_tile(p, first + (q / chunks) * p.grid, TROWS);
            float* ep = (float*)(img + XS_WORDS + WS_WORDS);
            const int m = tp.mt * TM + pt;
            const float c0 = (pp.oscale ? pp.oscale[(size_t)tp.n * p.m + m] : 1.f) * pp.gain;
            const float c1 = (pp.bias ? pp.bias[m] : 0.f) * pp.gain;
            const float al = pp.act == 3 ? pp.alpha : 1.f;
            ep[pt] = c0;
            ep[TM + pt] = c1;
            ep[2 * TM + pt] = c0 * al;
            ep[3 * TM + pt] = c1 * al;
        };

        load_x(0);
        dma_w(0, lds);
        store_x(lds);
        put_ep(0, lds);
        if (total > 1) load_x(1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // image 0 ready
        for (int q = 0; q < total; q++) {
            if (q + 1 < total) {
                u32x4* img = lds + ((q + 1) & 1) * WS_IMAGE_WORDS;    // last read by the consumers in iteration q-1, i.e. before the previous barrier
                store_x(img);
                put_ep(q + 1, img);
                dma_w(q + 1, img);
                if (q + 2 < total) load_x(q + 2);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // =========================================== consumers ===========================================
    const int l32 = lane & 31, g = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][hf][e] = 0.f;


    __builtin_amdgcn_s_barrier();   // image 0 ready
    asm volatile("" ::: "memory");
    for (int q = 0; q < total; q++) {
        const u32x4* xs = lds + (q & 1) * WS_IMAGE_WORDS;
        const u32x4* ws = xs + XS_WORDS;
        const int c = q % chunks;
        // per-lane operand positions, recomputed per chunk from an opaque copy of the lane id: kept live across the loop they get
        // spilled to scratch, and the reload's `s_waitcnt vmcnt(0)` would also wait for the previous tile's 128 stores
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int a_lane = (ln >> 5) * TM + (ln & 31);                              // + ((hl * 9 + tap) * 2) * TM + hf * 32
        const int b_lane = ((ln >> 5) * RIN + 4 * wave) * PIN + (ln & 31);          // + (r + ky) * PIN + kx (+ 2 * XS_PLANE for lo)

        // 18 half-taps (tap, row pair); the operands of step s+1 are fetched before the 12 MFMAs of step s are issued
        u32x4 a[2][2][2];    // [buffer][half][hl]
        u32x4 b[2][2][2];    // [buffer][row][hl]
        auto fetch_a = [&](int buf, int tap) {
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                a[buf][hf][0] = ws[a_lane + ((0 * 9 + tap) * 2) * TM + hf * 32];
                if (TERMS > 1) a[buf][hf][1] = ws[a_lane + ((1 * 9 + tap) * 2) * TM + hf * 32];
            }
        };
        auto fetch_b = [&](int buf, int tap, int rh) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int pos = b_lane + (2 * rh + r + ky) * PIN + kx;
                b[buf][r][0] = xs[pos];
                if (TERMS > 1) b[buf][r][1] = xs[2 * XS_PLANE + pos];
            }
        };
        fetch_a(0, 0);
        fetch_b(0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TERMS > 1 ? 8 : 4, 0);   // the operands of step 0 come first, as one group
#pragma unroll
        for (int s = 0; s < 18; s++) {
            const int tap = s >> 1, rh = s & 1;
            const int ab = tap & 1, bb = s & 1;
            // operands of the next row pair, and -- a whole step ahead of their first use -- the weights of the next tap
            if (s + 1 < 18) fetch_b(bb ^ 1, (s + 1) >> 1, (s + 1) & 1);
            if (rh == 0 && tap + 1 < 9) fetch_a(ab ^ 1, tap + 1);
            if (TERMS > 1) {
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[2 * rh + r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ab][hf][1]), __builtin_bit_cast(bf16x8, b[bb][r][0]), acc[2 * rh + r][hf], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[2 * rh + r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ab][hf][0]), __builtin_bit_cast(bf16x8, b[bb][r][1]), acc[2 * rh + r][hf], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    acc[2 * rh + r][hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ab][hf][0]), __builtin_bit_cast(bf16x8, b[bb][r][0]), acc[2 * rh + r][hf], 0, 0, 0);
            // pin the interleave: one operand read of step s+1 behind each of the first MFMAs of step s (the MFMA issues every 32 cycles,
            // a ds_read_b128 costs one issue slot), so the reads are spread over the step and nothing is fetched earlier than needed
            constexpr int MF = TERMS > 1 ? 12 : 4;
            const int reads = (TERMS > 1 ? 2 : 1) * ((s + 1 < 18 ? 2 : 0) + (rh == 0 && tap + 1 < 9 ? 2 : 0));
#pragma unroll
            for (int i = 0; i < MF; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // 1 MFMA
                if (i < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
            }
        }

        if (c == chunks - 1) {
            // C layout: col (pixel) = lane & 31, row (m) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5): 128-B contiguous stores
            const tile_pos tp = decode_tile(p, first + (q / chunks) * p.grid, TROWS);
            float* yb = p.y + ((size_t)tp.n * p.m + tp.mt * TM) * plane + (size_t)(tp.y0 + 4 * wave) * p.w + tp.x0 + l32;
            const float* ep = (const float*)(ws + WS_WORDS);
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    const int m0 = hf * 32 + 8 * e4 + 4 * g;
                    f32x4 c0, c1, c2, c3;
                    if (EPI == 1) { c0 = *(const f32x4*)(ep + m0); c1 = *(const f32x4*)(ep + TM + m0); c2 = *(const f32x4*)(ep + 2 * TM + m0); c3 = *(const f32x4*)(ep + 3 * TM + m0); }
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int ei = 0; ei < 4; ei++) {
                            float v = acc[r][hf][4 * e4 + ei];
                            if (EPI == 1) {
                                v = fmaxf(__builtin_fmaf(v, c0[ei], c1[ei]), __builtin_fmaf(v, c2[ei], c3[ei]));
                                if (pp.clamp >= 0.f) v = (v > -pp.clamp & v < pp.clamp) ? v : (v >= 0.f) ? pp.clamp : -pp.clamp;
                            }
                            yb[(size_t)(m0 + ei) * plane + (size_t)r * p.w] = v;
                            acc[r][hf][4 * e4 + ei] = 0.f;
                        }
                }
        }
        // every LDS read of this image has been consumed by an MFMA above; the asm statements keep the compiler from moving LDS
        // accesses across the barrier (a plain s_barrier is not a memory fence, and __syncthreads() would also drain the stores)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

}  // namespace sgv_conv
