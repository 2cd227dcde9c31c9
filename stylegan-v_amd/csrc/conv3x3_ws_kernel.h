// 3x3 / stride 1 / pad 1 convolution on NCHW fp32 tensors -- producer / consumer form of conv3x3_kernel (conv3x3_kernel.h), same
// arithmetic (TERMS = 4, the default: block-scaled 2-way fp16 split, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate -- csrc/sgv_split.h;
// TERMS = 3: the bf16 hi/lo split of rounds 1-3 on v_mfma_f32_32x32x16_bf16; TERMS = 1: one 16-bit product), same LDS images, same weight layout.
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
//   waves 4-6 (x producers): start the loads of chunk q+2 into one of two register sets, then write chunk q+1 (loaded an iteration
//                          ago) into image (q+1) & 1 -- optional per-(sample, channel) scale, hi/lo split, transposing ds_write_b128;
//   wave 7 (weight DMA):   the weight block of chunk q+1 as global -> LDS DMA (36 x 1 KiB `global_load_lds_dwordx4`: no registers,
//                          no ds_write), waited for before the barrier.
//
// One barrier per chunk.  The producers' VALU / LDS-write / VMEM instructions issue in the gaps between the consumer's MFMAs of the
// same SIMD (separate pipes).  LDS: 2 x (x tile 38.3 KiB + weights 36 KiB + 512 B epilogue vectors) = 149.5 KiB, one workgroup per CU.
#pragma once
#include <type_traits>

#include "conv3x3_kernel.h"
#include "sgv_io16.h"

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
    int accumulate;        // y += result (one no-return fp32 atomic per element) instead of y = result
    float* y_amax;         // fp32 tensors: max |stored value| as a by-product (sgv_amax_sink; with `accumulate`: of the INCREMENT), or NULL
    unsigned long long* stamp;   // NULL, or {start, end} in the 100-MHz device clock, written by the kernel itself (sgv_launch_scope::kernel_stamps: per-launch timing inside a
                                 // replayed hipGraph without bracketing kernels): start = workgroup 0 as it begins, end = atomicMax over the consumer waves as they leave
};

// PRO: 0 plain x, 1 x * xscale[n,k].
// EPI: 0 plain store; 1: y = clamp(lrelu_alpha(acc * oscale[n,m] + bias[m]) * gain), evaluated as max(fma(acc, c0, c1), fma(acc, c2, c3)) with
//      c0 = oscale * gain, c1 = bias * gain, c2 = c0 * alpha, c3 = c1 * alpha prepared per tile by the producers (valid for gain > 0,
//      0 <= alpha <= 1; alpha = 1 is the linear activation): 3 VALU operations per output in the consumer, which is what the MFMA waves
//      can afford.  Differs from the three-pass composition (networks.py:70-71 + bias_act.cu:39-146) by fused-multiply-add rounding only.
template <int ABL, int TERMS = 3>
__device__ __forceinline__ f32x16 ws_mma(u32x4 a, u32x4 b, f32x16 c) {
    if (ABL == 2) { asm volatile("" :: "v"(a), "v"(b)); return c; }   // lab ablation: keep the operand reads, drop the MFMA
    return mma16<TERMS>(a, b, c);      // bf16 operands; fp16 ones for the block-scaled split (TERMS = 4, sgv_split.h)
}

// ABL (tools/conv_lab.hip only; results are wrong by construction): 1 producers only keep the barrier protocol (consumer-only speed), 2 no MFMAs
// (operand reads + barriers), 3 no operand reads (MFMA issue + barriers), 4 no epilogue stores, 5 producers without global loads / DMA.
// PRIO: s_setprio level of the consumer waves (the producers' VALU-heavy split competes for the SIMD's issue slots).
// IO: element format of x and y (sgv_io16.h): 0 fp32; 1 bf16 / 2 fp16 tensors (TERMS = 1: one bf16 operand per value, fp32 accumulate, 16-bit stores;
//     weights, scales and the bias stay fp32) -- the mixed-precision blocks of the reference (networks.py:227,461), same loads count, same waits.
//      6 = 1 + 4, 7 = 1 + 3 (consumer loop alone, without its stores / without its operand reads); 8 / 9: as 5, but only the weight DMA / only the x loads are dropped.
// ORD: order of the consumers' 216 MFMAs per chunk.  0: tap-major (for every tap its four rows: 12 operand reads per 24 MFMAs).  1: column-major with ROW
//      REUSE -- for kx = 0..2 the six input rows j = 0..5 of the wave are read once each and serve every (output row r, ky) with r + ky = j
//      (6 / 12 / 18 / 18 / 12 / 6 MFMAs per row), the three weight taps (ky, kx) of the column stay in registers and are refilled for the next column as
//      their last use passes: 72 instead of 108 ds_read_b128 per chunk.  The consumer loop is LDS-read bound (profiles/r03_conv_lab_ws_ablations.log: the loop alone
//      0.81 ms with its operand reads, 0.60 ms without them, on every layer shape) -- but NOT by their number: measured equal or 1-4 % slower than the
//      tap-major order on every shape (profiles/r03_conv_lab_ws_row_reuse.log; consumers alone 0.806 vs 0.808 ms).  The 0.60 ms of the no-read ablation is the
//      matrix pipe on constant operands (no data toggling, 2.2 GHz); on real data the part holds ~1.85 GHz and the loop alone is at 89 % of that ceiling.
//      Kept as a lab variant (default 0); same products, a different summation order (kx outermost) than the 4-wave kernel.
// TERMS: 1 bf16 products, 3 bf16 split, 4 block-scaled fp16 split (fp32-grade; sgv_split.h) -- same LDS images, same MFMA count as 3.
template <int TERMS, int PRO, int EPI, int ABL = 0, int PRIO = 1, int IO = 0, int ORD = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(conv_ws_params pp) {
    constexpr int F = sgv_conv::operand_format<TERMS, IO>();      // operand format of the products (sgv_split.h): TERMS, or 2 = fp16 operands for fp16 tensors
    constexpr bool A_IDLE = ABL == 1 || ABL == 6 || ABL == 7, A_NOSTORE = ABL == 4 || ABL == 6, A_NOREAD = ABL == 3 || ABL == 7;
    static_assert(IO == 0 || TERMS == 1, "16-bit tensors are multiplied as single bf16 operands");
    using namespace sgv_io;
    const conv_params& p = pp.c;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int chunks = p.k / KC;
    const size_t plane = (size_t)p.h * p.w;
    // TERMS = 4: the operands' block exponents (wave-uniform; read before any counted asm load is in flight)
    const int ex = operand_exponent<TERMS>(p.x_amax, p.x_amax2), ew = operand_exponent<TERMS>(p.w_amax);
    const float xS = split_scale(ex);
    const int eu = unscale_exponent(ex, ew);

    // flat sequence of (tile, chunk) pairs of this workgroup: q -> tile = first + (q / chunks) * grid, chunk = q % chunks
    const int first = xcd_swizzle(blockIdx.x, gridDim.x);
    if (first >= p.tiles) return;
    const int my_tiles = (p.tiles - first + p.grid - 1) / p.grid;
    const int total = my_tiles * chunks;

    if (wave == 7) {
        if (pp.stamp && blockIdx.x == 0 && lane == 0) *pp.stamp = wall_clock64();     // (this wave only ever waits for vmcnt(0))
        // =========================================== weight DMA wave ===========================================
        // The 36-KiB weight block of (tile, chunk) q is copied verbatim: 36 wave-instructions of 64 lanes x 16 B to a wave-uniform LDS base.
        // A wave of its own, so that the x waves' instruction stream contains no LDS-DMA (hipcc drains vmcnt to 0 at every use of an
        // ordinary load while one is in flight, which would serialise their prefetch).
        auto dma_w = [&](int q, u32x4* img) {
            const tile_pos tp = decode_tile(p, first + (q / chunks) * p.grid, TROWS);
            const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + (q % chunks)) * WS_WORDS + lane;
            u32x4* wl = img + XS_WORDS;
#pragma unroll
            for (int j = 0; j < 36; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wq + j * 64),
                                                 (__attribute__((address_space(3))) void*)(wl + j * 64), 16, 0, 0);
        };
        dma_w(0, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // image 0 ready
        for (int q = 0; q < total; q++) {
            if (q + 1 < total && !A_IDLE && ABL != 5 && ABL != 8) dma_w(q + 1, lds + ((q + 1) & 1) * WS_IMAGE_WORDS);   // free since the previous barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    if (wave >= 4) {
        // =========================================== x waves (4, 5, 6) ===========================================
        // 360 units per chunk: 288 items of 8 channels x 4 pixels (rows 0..17 of the tile incl. its row halo) and 72 column-halo items of
        // 8 channels x 1 pixel.  Thread i of 192 takes units i and i + 192.  Two register sets: the loads of chunk q+2 are issued at the
        // start of iteration q and consumed at the start of iteration q+1 -- a whole consumer iteration to land.
        const int pt = t - 256;
        constexpr int ITEMS = 16 * RIN;
        const int u1 = pt + 192;
        const int kind1 = u1 < ITEMS ? 0 : (u1 < ITEMS + 4 * RIN ? 1 : 2);        // second unit: item, column halo, none
        const int oct = pt & 1;                                                   // same for both units (192 and ITEMS are even)
        const int a_quad = (pt >> 1) & 7, a_row0 = pt >> 4, a_row1 = u1 >> 4;     // u1 >> 4 = a_row0 + 12
        const int hh = u1 - ITEMS, h_side = (hh >> 1) & 1, h_row = hh >> 2;
        struct xset { px4<IO> a[8]; px4<IO> b[8]; f32x4 sc[2]; float ep0, ep1; bool ok0, ok1; };

        // Branch-free: every thread issues the same 16 (+2) loads per chunk -- out-of-image positions load from a clamped address and are zeroed
        // when they are written to LDS -- so that the compiler can count them (`s_waitcnt vmcnt(16)` before the previous set is consumed)
        // instead of draining to 0.  A column-halo pixel is taken from the aligned 16-B group that contains it.
        auto load_x = [&](int q, xset& r) {
            const tile_pos tp = decode_tile(p, first + (q / chunks) * p.grid, TROWS);
            const int c = q % chunks;
            const size_t xb_ = ((size_t)tp.n * p.k + c * KC + 8 * oct) * plane;      // element offsets into p.x
            const int gy0 = tp.y0 - 1 + a_row0;
            const int gy1 = tp.y0 - 1 + (kind1 == 0 ? a_row1 : h_row);
            const int gx0 = tp.x0 + 4 * a_quad;
            const int gx1 = kind1 == 0 ? gx0 : (h_side ? tp.x0 + SEG : tp.x0 - 4);
            r.ok0 = gy0 >= 0 && gy0 < p.h;
            r.ok1 = kind1 != 2 && gy1 >= 0 && gy1 < p.h && gx1 >= 0 && gx1 < p.w;
            const size_t q0 = xb_ + (size_t)min(max(gy0, 0), p.h - 1) * p.w + gx0;
            const size_t q1 = xb_ + (size_t)min(max(gy1, 0), p.h - 1) * p.w + min(max(gx1, 0), p.w - 4);
            // The loads are inline asm on purpose: hipcc's own wait insertion cannot keep a load in flight across the loop back edge (it
            // emitted vmcnt(15) where 31 was needed, i.e. waited for one of the loads just issued), which puts the HBM latency back on
            // the producers' path.  Their destinations are unprotected until the counted s_waitcnt in `arrive` (cdna_hip_programming.md 5.7).
#pragma unroll
            for (int j = 0; j < 8; j++) px4_load<IO>(r.a[j], at<IO>(p.x, q0 + j * plane));
#pragma unroll
            for (int j = 0; j < 8; j++) px4_load<IO>(r.b[j], at<IO>(p.x, q1 + j * plane));
            if (PRO == 1) {
                const float* sp = pp.xscale + (size_t)tp.n * p.k + c * KC + 8 * oct;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.sc[0]) : "v"(sp) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.sc[1]) : "v"(sp + 4) : "memory");
            }
            if (EPI >= 1) {
                // the tile's output scale and bias (consumed by put_ep with the tile's LAST chunk) travel with every chunk's set as two more counted
                // loads: as plain C++ loads inside put_ep the compiler put `s_waitcnt vmcnt(0)` in front of their use, which drained the chunk
                // prefetch queue once per tile (64-channel layers: 4 chunks per tile)
                const int m = min(tp.mt * TM + (pt & (TM - 1)), p.m - 1);     // (a half-full last m tile: its upper rows are never stored)
                const float* po = pp.oscale ? pp.oscale + (size_t)tp.n * p.m + m : (const float*)p.x;
                const float* pb = pp.bias ? pp.bias + m : (const float*)p.x;
                asm volatile("global_load_dword %0, %1, off" : "=v"(r.ep0) : "v"(po) : "memory");
                asm volatile("global_load_dword %0, %1, off" : "=v"(r.ep1) : "v"(pb) : "memory");
            }
        };
        // everything issued before the `newer` most recent loads has landed; re-define the set's registers after the wait so that no use can be
        // scheduled above it
        auto arrive = [&](xset& r, bool newer) {
            if (newer) {   // loads per set: 16 + 2 (PRO) + 2 (EPI)
                if (PRO == 1 && EPI >= 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else if (PRO == 1 || EPI >= 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) { px4_pin<IO>(r.a[j]); px4_pin<IO>(r.b[j]); }
            if (PRO == 1) { asm volatile("" : "+v"(r.sc[0])); asm volatile("" : "+v"(r.sc[1])); }
            if (EPI >= 1) { asm volatile("" : "+v"(r.ep0)); asm volatile("" : "+v"(r.ep1)); }
        };
        auto put = [&](u32x4* xs, int pos, float* v, const xset& r, bool ok) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = ok ? (PRO == 1 ? v[j] * r.sc[j >> 2][j & 3] : v[j]) : 0.f;
            u32x4 hi, lo;
            split8t<F>(v, xS, hi, lo);
            xs[pos] = hi;
            if (TERMS > 1) xs[2 * XS_PLANE + pos] = lo;
        };
        auto store_x = [&](u32x4* xs, const xset& r) {
            {
                const int base = (oct * RIN + a_row0) * PIN + 1 + 4 * a_quad;
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = px4_get<IO>(r.a[j], px);
                    put(xs, base + px, v, r, r.ok0);
                }
            }
            if (kind1 == 0) {
                const int base = (oct * RIN + a_row1) * PIN + 1 + 4 * a_quad;
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = px4_get<IO>(r.b[j], px);
                    put(xs, base + px, v, r, r.ok1);
                }
            } else if (kind1 == 1) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = h_side ? px4_get<IO>(r.b[j], 0) : px4_get<IO>(r.b[j], 3);   // right halo: first pixel of the next group; left halo: last of the previous
                put(xs, (oct * RIN + h_row) * PIN + (h_side ? SEG + 1 : 0), v, r, r.ok1);
            }
        };
        auto put_ep = [&](int q, u32x4* img, const xset& r) {   // the tile's epilogue vectors ride with its LAST chunk (r: that chunk's set)
            if (EPI == 0 || (q % chunks) != chunks - 1 || pt >= TM) return;
            float* ep = (float*)(img + XS_WORDS + WS_WORDS);
            float c0 = (pp.oscale ? r.ep0 : 1.f) * pp.gain;
            if (TERMS == 4) c0 = __builtin_ldexpf(c0, eu);      // the accumulators carry both operands' block scales
            const float c1 = (pp.bias ? r.ep1 : 0.f) * pp.gain;
            const float al = pp.act == 3 ? pp.alpha : 1.f;
            ep[pt] = c0;
            ep[TM + pt] = c1;
            ep[2 * TM + pt] = c0 * al;
            ep[3 * TM + pt] = c1 * al;
        };
        // iteration q: start the loads of chunk q+2 into `ld`, then split / write chunk q+1 (in `st`, loaded one iteration ago) into image (q+1) & 1
        auto step = [&](int q, xset& ld, xset& st) {
            const bool more = q + 2 < total && !A_IDLE && ABL != 5 && ABL != 9;
            if (more) load_x(q + 2, ld);
            if (q + 1 < total && !A_IDLE) {
                u32x4* img = lds + ((q + 1) & 1) * WS_IMAGE_WORDS;    // last read by the consumers in iteration q-1, i.e. before the previous barrier
                arrive(st, more);
                store_x(img, st);
                put_ep(q + 1, img, st);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS writes; the loads just issued stay in flight across the barrier
            __builtin_amdgcn_s_barrier();
        };

        xset s0, s1;
        load_x(0, s0);
        if (total > 1 && !A_IDLE) load_x(1, s1);     // (lab: a set that is never consumed must not be loaded -- its registers are unprotected)
        arrive(s0, total > 1 && !A_IDLE);
        store_x(lds, s0);
        put_ep(0, lds, s0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // image 0 ready
        for (int q = 0; q < total; q += 2) {
            step(q, s0, s1);
            if (q + 1 < total) step(q + 1, s1, s0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing is outstanding here on the product path; the lab's ablations skip consumers of issued loads
        return;
    }

    // =========================================== consumers ===========================================
    f32x16 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][hf][e] = 0.f;


    if (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
    __builtin_amdgcn_s_barrier();   // image 0 ready
    asm volatile("" ::: "memory");
    unsigned amx = 0;               // max |stored value| of this lane's tiles (bit pattern; IO == 0 only)
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

        if (ORD == 1) {
            // ---- column-major with row reuse: 18 row steps (kx, j); A[ky] = the weights of tap (ky, kx) for both m halves, B = input row j at column offset kx ----
            u32x4 A[3][2][2];    // [ky][half][hl]
            u32x4 B[2][2];       // [buffer][hl]
            auto fetch_A = [&](int ky, int kx) {
                const int tap = ky * 3 + kx;
                if (A_NOREAD) { for (int hf = 0; hf < 2; hf++) for (int hl = 0; hl < 2; hl++) A[ky][hf][hl] = u32x4{(unsigned)ln, 2u, 3u, 4u}; return; }
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    A[ky][hf][0] = ws[a_lane + ((0 * 9 + tap) * 2) * TM + hf * 32];
                    if (TERMS > 1) A[ky][hf][1] = ws[a_lane + ((1 * 9 + tap) * 2) * TM + hf * 32];
                }
            };
            auto fetch_B = [&](int buf, int j, int kx) {
                if (A_NOREAD) { for (int hl = 0; hl < 2; hl++) B[buf][hl] = u32x4{5u, (unsigned)ln, 7u, 8u}; return; }
                const int pos = b_lane + j * PIN + kx;
                B[buf][0] = xs[pos];
                if (TERMS > 1) B[buf][1] = xs[2 * XS_PLANE + pos];
            };
            constexpr int RA = TERMS > 1 ? 4 : 2, RB = TERMS > 1 ? 2 : 1;   // ds_read_b128 per weight tap / per input row
            fetch_B(0, 0, 0);
            fetch_A(0, 0);
            fetch_A(1, 0);
            fetch_A(2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RB + 3 * RA, 0);   // the operands of the first rows come first, as one group
#pragma unroll
            for (int s = 0; s < 18; s++) {
                const int kx = s / 6, j = s % 6, bb = s & 1;
                int reads = 0;
                if (s + 1 < 18) { fetch_B(bb ^ 1, (s + 1) % 6, (s + 1) / 6); reads += RB; }
                // the tap whose last use was the previous row step is refilled for the next column: ky = 0 after j = 3, ky = 1 after j = 4, ky = 2 after j = 5
                if (j == 4 && kx < 2) { fetch_A(0, kx + 1); reads += RA; }
                if (j == 5 && kx < 2) { fetch_A(1, kx + 1); reads += RA; }
                if (j == 0 && kx > 0) { fetch_A(2, kx); reads += RA; }
                const int ky_lo = j > 3 ? j - 3 : 0, ky_hi = j < 2 ? j : 2;     // (r, ky) with r = j - ky in 0..3
                if (TERMS > 1) {
#pragma unroll
                    for (int ky = ky_lo; ky <= ky_hi; ky++)
#pragma unroll
                        for (int hf = 0; hf < 2; hf++)
                            acc[j - ky][hf] = ws_mma<ABL, F>(A[ky][hf][1], B[bb][0], acc[j - ky][hf]);
#pragma unroll
                    for (int ky = ky_lo; ky <= ky_hi; ky++)
#pragma unroll
                        for (int hf = 0; hf < 2; hf++)
                            acc[j - ky][hf] = ws_mma<ABL, F>(A[ky][hf][0], B[bb][1], acc[j - ky][hf]);
                }
#pragma unroll
                for (int ky = ky_lo; ky <= ky_hi; ky++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[j - ky][hf] = ws_mma<ABL, F>(A[ky][hf][0], B[bb][0], acc[j - ky][hf]);
                const int MF = (TERMS > 1 ? 6 : 2) * (ky_hi - ky_lo + 1);
#pragma unroll
                for (int i = 0; i < MF; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // 1 MFMA
                    if (i < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
                }
                if (reads > MF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (reads > MF + 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
        // 18 half-taps (tap, row pair); the operands of step s+1 are fetched before the 12 MFMAs of step s are issued
        u32x4 a[2][2][2];    // [buffer][half][hl]
        u32x4 b[2][2][2];    // [buffer][row][hl]
        auto fetch_a = [&](int buf, int tap) {
            if (A_NOREAD) { for (int hf = 0; hf < 2; hf++) for (int hl = 0; hl < 2; hl++) a[buf][hf][hl] = u32x4{(unsigned)ln, 2u, 3u, 4u}; return; }
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                a[buf][hf][0] = ws[a_lane + ((0 * 9 + tap) * 2) * TM + hf * 32];
                if (TERMS > 1) a[buf][hf][1] = ws[a_lane + ((1 * 9 + tap) * 2) * TM + hf * 32];
            }
        };
        auto fetch_b = [&](int buf, int tap, int rh) {
            if (A_NOREAD) { for (int r = 0; r < 2; r++) for (int hl = 0; hl < 2; hl++) b[buf][r][hl] = u32x4{5u, (unsigned)ln, 7u, 8u}; return; }
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
                        acc[2 * rh + r][hf] = ws_mma<ABL, F>(a[ab][hf][1], b[bb][r][0], acc[2 * rh + r][hf]);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
                        acc[2 * rh + r][hf] = ws_mma<ABL, F>(a[ab][hf][0], b[bb][r][1], acc[2 * rh + r][hf]);
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    acc[2 * rh + r][hf] = ws_mma<ABL, F>(a[ab][hf][0], b[bb][r][0], acc[2 * rh + r][hf]);
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
        }

        if (c == chunks - 1) {
            // C layout: col (pixel) = lane & 31, row (m) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5): 128-B contiguous stores
            const tile_pos tp = decode_tile(p, first + (q / chunks) * p.grid, TROWS);
            int le = lane;                      // opaque copy: everything lane-dependent below is recomputed per tile instead of being
            asm volatile("" : "+v"(le));        // hoisted out of the chunk loop and spilled (nine scratch reloads, each behind a vmcnt(0))
            const int g = le >> 5;
            const size_t yoff = ((size_t)tp.n * p.m + tp.mt * TM) * plane + (size_t)(tp.y0 + 4 * wave) * p.w + tp.x0 + (le & 31);
            float* yb = (float*)p.y + yoff;      // (fp32 tensors; 16-bit ones are stored through out_store)
            const float* ep = (const float*)(ws + WS_WORDS);
            // The clamp is one v_med3_f32 -- which returns min3 when an operand is NaN, so a NaN accumulator would be stored as -clamp.  With a
            // clamp that IS bias_act's result (bias_act.cu:142: every comparison with NaN fails -> -clamp); without one the reference propagates the
            // NaN, so the un-clamped store path has no med3 at all: two instantiations of the store loop behind one wave-uniform branch per tile.
            auto store_tile = [&](auto clamped) {
                constexpr bool CLAMP = decltype(clamped)::value;
                const float clamp_hi = pp.clamp;
                const int m_left = p.m - tp.mt * TM;     // output channels of this tile that exist: 64, or 32 in a half-full last tile (wave-uniform)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int e4 = 0; e4 < 4; e4++) {
                        if (hf * 32 >= m_left) {     // the padded half: zero weights in, nothing out
#pragma unroll
                            for (int r = 0; r < 4; r++)
#pragma unroll
                                for (int ei = 0; ei < 4; ei++) acc[r][hf][4 * e4 + ei] = 0.f;
                            continue;
                        }
                        const int m0 = hf * 32 + 8 * e4 + 4 * g;
                        f32x4 c0, c1, c2, c3;
                        if (EPI == 1) { c0 = *(const f32x4*)(ep + m0); c1 = *(const f32x4*)(ep + TM + m0); c2 = *(const f32x4*)(ep + 2 * TM + m0); c3 = *(const f32x4*)(ep + 3 * TM + m0); }
#pragma unroll
                        for (int r = 0; r < 4; r++)
#pragma unroll
                            for (int ei = 0; ei < 4; ei++) {
                                float v = acc[r][hf][4 * e4 + ei];
                                if (TERMS == 4 && EPI != 1) v = __builtin_ldexpf(v, eu);
                                if (EPI == 1) {
                                    v = fmaxf(__builtin_fmaf(v, c0[ei], c1[ei]), __builtin_fmaf(v, c2[ei], c3[ei]));
                                    if (CLAMP) v = __builtin_amdgcn_fmed3f(v, -clamp_hi, clamp_hi);
                                }
                                if (IO == 0) amx = sgv_amax_fold(amx, v);
                                if (A_NOSTORE) asm volatile("" :: "v"(v));
                                else if (IO != 0) out_store<IO>(p.y, yoff + (size_t)(m0 + ei) * plane + (size_t)r * p.w, v);
                                else if (pp.accumulate) atomicAdd(yb + (size_t)(m0 + ei) * plane + (size_t)r * p.w, v);
                                else yb[(size_t)(m0 + ei) * plane + (size_t)r * p.w] = v;
                                acc[r][hf][4 * e4 + ei] = 0.f;
                            }
                    }
            };
            if (EPI == 1 && pp.clamp >= 0.f) store_tile(std::true_type{}); else store_tile(std::false_type{});
        }
        // every LDS read of this image has been consumed by an MFMA above; the asm statements keep the compiler from moving LDS
        // accesses across the barrier (a plain s_barrier is not a memory fence, and __syncthreads() would also drain the stores)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (IO == 0 && pp.y_amax) sgv_amax_commit(amx, pp.y_amax);     // (the four consumer waves; whole waves)
    if (pp.stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the tile's stores have left
        if (lane == 0) atomicMax(pp.stamp + 1, (unsigned long long)wall_clock64());      // the clock never runs backwards: no reset between replays
    }
}

}  // namespace sgv_conv
