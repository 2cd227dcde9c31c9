// Producer / consumer forms of the stride-2 members of the 3x3 family (conv3x3s2_kernel.h holds the one-role-per-wave forms and the
// arithmetic: bf16 hi/lo split products on v_mfma_f32_32x32x16_bf16, fp32 accumulate; same prepared weights).
//
//   conv3x3_s2_ws_kernel   y[n,m,Y,X] = sum_{k,ky,kx} wgt(m,k,ky,kx) * x[n,k,2Y+ky,2X+kx]        x: (2H+1)x(2W+1) -> y: HxW
//
// Reference: the `conv2d(stride=2)` behind the FIR of the down-sampling path (src/torch_utils/ops/conv2d_resample.py:113-126) and
// the data gradient of the up-sampling path's `conv_transpose2d` (conv2d_gradfix.py:100-118).
//
// Why: in conv3x3_s2_kernel one workgroup (108 KiB LDS) owns a CU with ONE wave per SIMD that loads, splits, writes LDS and issues
// the MFMAs in turn -- the matrix pipe idles 70 % of the time (profiles/r02_pmc_bench_step_MFMA_table.txt).  Same cure as
// conv3x3_ws_kernel.h: 8 waves, roles split, LDS double-buffered, one barrier per 16-channel chunk.
//
//   waves 0-3 (consumers): wave = one output row x 32 px x 64 m (2 accumulator tiles); per tap 6 MFMAs against 6 operand reads that
//                          were fetched one tap ahead;
//   waves 4-6 (x producers): a stride-2 tile needs 4x the input of a stride-1 tile per MFMA, so the tile is 4 output rows: 16 channels x
//                          9 rows x 65 columns.  Rows of the (2W+1)-wide tensor are only 4-byte aligned: unaligned `global_load_dwordx4`
//                          (gfx950 runs with unaligned access mode on), items of 8 channels x 4 columns, de-interleaved into an even and an
//                          odd column plane on their way into LDS so that tap kx reads plane kx & 1 at pixel + (kx >> 1);
//   wave 7 (weight DMA):   36 KiB per chunk, global -> LDS.
//
// LDS: 2 x (x 37.1 KiB + weights 36 KiB) = 146.3 KiB, one workgroup per CU.
#pragma once
#include <type_traits>

#include "conv3x3s2_kernel.h"
#include "conv3x3_ws_kernel.h"
#include "sgv_io16.h"

namespace sgv_conv {

constexpr int S2W_ROWS = 4;                              // output rows per tile
constexpr int S2W_RIN = 2 * S2W_ROWS + 1;                // 9 input rows
constexpr int S2W_PW = 33;                               // words per parity plane row (the even plane holds 33 columns)
constexpr int S2W_XS_PLANE = S2W_RIN * 2 * S2W_PW;       // words per (hl, octet)
constexpr int S2W_XS_WORDS = 4 * S2W_XS_PLANE;
constexpr int S2W_IMAGE_WORDS = S2W_XS_WORDS + WS_WORDS;
constexpr int S2W_LDS_BYTES = 2 * S2W_IMAGE_WORDS * 16;

// ABL (tools/conv_s2_lab.hip only; results are wrong by construction): 1 x loads from 16-byte aligned addresses, 2 no weight DMA, 3 no x loads,
// 4 no MFMAs, 5 no x split / LDS writes, 6 consumers only keep the barrier protocol (producer + DMA pipeline speed),
// 7 producers and DMA only keep the barrier protocol (consumer speed).
template <int TERMS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_s2_ws_kernel(s2_params p) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int chunks = p.k / KC;
    const int win = 2 * p.w + 1;
    const size_t plane_in = (size_t)(2 * p.h + 1) * win, plane_out = (size_t)p.h * p.w;
    const int ex = operand_exponent<TERMS>(p.x_amax), ew = operand_exponent<TERMS>(p.w_amax);   // TERMS = 4: block exponents (sgv_split.h)
    const float xS = split_scale(ex);
    const int eu = unscale_exponent(ex, ew);

    // order 0: tile i of the launch goes to workgroup i % grid (the m tiles of one x tile run side by side on neighbouring CUs of an XCD);
    // order 1: a workgroup takes a spatial tile and runs all its m tiles back to back (x re-read by the same CU: L2-resident by construction)
    const int mts = p.m / TM;
    const int first = xcd_swizzle(blockIdx.x, gridDim.x);
    const int units = p.order ? p.tiles / mts : p.tiles;
    if (first >= units) return;
    const int my_units = (units - first + p.grid - 1) / p.grid;
    const int total = (p.order ? my_units * mts : my_units) * chunks;
    auto tile_of = [&](int q) {
        const int j = q / chunks;
        return p.order ? (first + (j / mts) * p.grid) * mts + j % mts : first + j * p.grid;
    };

    if (wave == 7) {
        // =========================================== weight DMA wave ===========================================
        auto dma_w = [&](int q, u32x4* img) {
            const tile_pos tp = decode_tile_s2(p, tile_of(q), S2W_ROWS);
            const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + (q % chunks)) * WS_WORDS + lane;
            u32x4* wl = img + S2W_XS_WORDS;
#pragma unroll
            for (int j = 0; j < 36; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wq + j * 64),
                                                 (__attribute__((address_space(3))) void*)(wl + j * 64), 16, 0, 0);
        };
        dma_w(0, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // image 0 ready
        for (int q = 0; q < total; q++) {
            if (q + 1 < total && ABL != 2 && (ABL != 7 && ABL != 10)) dma_w(q + 1, lds + ((q + 1) & 1) * S2W_IMAGE_WORDS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    if (wave >= 4) {
        // =========================================== x waves (4, 5, 6) ===========================================
        // 306 units per chunk: 288 items of 8 channels x 4 columns (9 rows x 16 groups x 2 octets: columns 0..63 of the tile) and 18 items for
        // column 64 (taken as the last element of the group that ends there, so that nothing is read beyond the row).  Thread i of 192 takes
        // units i and i + 192; every thread issues the same 16 loads per chunk (idle second units load a valid address and drop the result).
        const int pt = t - 256;
        constexpr int ITEMS = 32 * S2W_RIN;
        const int u1 = pt + 192;
        const int kind1 = u1 < ITEMS ? 0 : (u1 < ITEMS + 2 * S2W_RIN ? 1 : 2);
        const int oct = pt & 1;
        const int a_grp = (pt >> 1) & 15, a_row0 = pt >> 5, a_row1 = u1 >> 5;
        const int h_row = (u1 - ITEMS) >> 1;
        struct xset { f32x4 a[8]; f32x4 b[8]; };

        auto load_x = [&](int q, xset& r) {
            const tile_pos tp = decode_tile_s2(p, tile_of(q), S2W_ROWS);
            const int c = q % chunks;
            const float* xb_ = p.x + ((size_t)tp.n * p.k + c * KC + 8 * oct) * plane_in + (size_t)(2 * tp.y0) * win + 2 * tp.x0;
            const float* q0 = xb_ + (size_t)a_row0 * win + 4 * a_grp;
            const float* q1 = xb_ + (kind1 == 0 ? (size_t)a_row1 * win + 4 * a_grp : kind1 == 1 ? (size_t)h_row * win + 61 : 0);
            if (ABL == 1) { q0 = (const float*)((uintptr_t)q0 & ~(uintptr_t)15); q1 = (const float*)((uintptr_t)q1 & ~(uintptr_t)15); }
            const size_t cstep = ABL == 1 ? (plane_in & ~(size_t)3) : plane_in;
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.a[j]) : "v"(q0 + j * cstep) : "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.b[j]) : "v"(q1 + j * cstep) : "memory");
        };
        auto arrive = [&](xset& r, bool newer) {
            if (newer) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("" : "+v"(r.a[j])); asm volatile("" : "+v"(r.b[j])); }
        };
        auto put = [&](u32x4* xs, int pos, const float* v) {
            u32x4 hi, lo;
            split8t<TERMS>(v, xS, hi, lo);
            xs[pos] = hi;
            if (TERMS > 1) xs[2 * S2W_XS_PLANE + pos] = lo;
        };
        // column 4 * grp + px of the tile -> plane (px & 1), word 2 * grp + (px >> 1)
        auto put_item = [&](u32x4* xs, int row, int grp, const f32x4* src) {
            const int base = (oct * S2W_RIN + row) * 2 * S2W_PW + 2 * grp;
#pragma unroll
            for (int px = 0; px < 4; px++) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = src[j][px];
                put(xs, base + (px & 1) * S2W_PW + (px >> 1), v);
            }
        };
        auto store_x = [&](u32x4* xs, const xset& r) {
            put_item(xs, a_row0, a_grp, r.a);
            if (kind1 == 0) put_item(xs, a_row1, a_grp, r.b);
            else if (kind1 == 1) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = r.b[j][3];
                put(xs, (oct * S2W_RIN + h_row) * 2 * S2W_PW + 32, v);
            }
        };
        auto step = [&](int q, xset& ld, xset& st) {
            const bool more = q + 2 < total && ABL != 3 && (ABL != 7 && ABL != 10);
            if (more) load_x(q + 2, ld);
            if (q + 1 < total) {
                arrive(st, more);
                if (ABL != 5 && (ABL != 7 && ABL != 10)) store_x(lds + ((q + 1) & 1) * S2W_IMAGE_WORDS, st);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

        xset s0, s1;
        load_x(0, s0);
        if (total > 1 && ABL != 3 && (ABL != 7 && ABL != 10)) load_x(1, s1);
        arrive(s0, total > 1 && ABL != 3 && (ABL != 7 && ABL != 10));
        store_x(lds, s0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // image 0 ready
        for (int q = 0; q < total; q += 2) {
            step(q, s0, s1);
            if (q + 1 < total) step(q + 1, s1, s0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =========================================== consumers ===========================================
    f32x16 acc[2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[hf][e] = 0.f;

    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_s_barrier();   // image 0 ready
    asm volatile("" ::: "memory");
    for (int q = 0; q < total; q++) {
        const u32x4* xs = lds + (q & 1) * S2W_IMAGE_WORDS;
        const u32x4* ws = xs + S2W_XS_WORDS;
        const int c = q % chunks;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int a_lane = (ln >> 5) * TM + (ln & 31);                                        // + ((hl * 9 + tap) * 2) * TM + hf * 32
        const int b_lane = ((ln >> 5) * S2W_RIN + 2 * wave) * 2 * S2W_PW + (ln & 31);         // + (ky * 2 + (kx & 1)) * PW + (kx >> 1)

        // operands are fetched TWO taps ahead (three register buffers): with one read behind each MFMA of a 6-MFMA tap, a distance of one tap
        // would leave the last reads a single MFMA (32 cycles) to land
        u32x4 a[3][2][2];    // [buffer][half][hl]
        u32x4 b[3][2];       // [buffer][hl]
        auto fetch = [&](int buf, int tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int pos = b_lane + (ky * 2 + (kx & 1)) * S2W_PW + (kx >> 1);
            // in the order of first use: a_lo x b_hi, a_hi x b_lo, a_hi x b_hi
            b[buf][0] = xs[pos];
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
                if (TERMS > 1) a[buf][hf][1] = ws[a_lane + ((1 * 9 + tap) * 2) * TM + hf * 32];
#pragma unroll
            for (int hf = 0; hf < 2; hf++) a[buf][hf][0] = ws[a_lane + ((0 * 9 + tap) * 2) * TM + hf * 32];
            if (TERMS > 1) b[buf][1] = xs[2 * S2W_XS_PLANE + pos];
        };
        constexpr int RD = TERMS > 1 ? 6 : 3;   // reads per tap
        if (ABL != 6) {
        fetch(0, 0);
        fetch(1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * RD, 0);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int bb = tap % 3;
            if (tap + 2 < 9) fetch((tap + 2) % 3, tap + 2);
            if (TERMS > 1) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++) acc[hf] = ws_mma<(ABL == 4 ? 2 : 0), TERMS>(a[bb][hf][1], b[bb][0], acc[hf]);
#pragma unroll
                for (int hf = 0; hf < 2; hf++) acc[hf] = ws_mma<(ABL == 4 ? 2 : 0), TERMS>(a[bb][hf][0], b[bb][1], acc[hf]);
            }
#pragma unroll
            for (int hf = 0; hf < 2; hf++) acc[hf] = ws_mma<(ABL == 4 ? 2 : 0), TERMS>(a[bb][hf][0], b[bb][0], acc[hf]);
            constexpr int MF = TERMS > 1 ? 6 : 2;
            const int reads = tap + 2 < 9 ? RD : 0;
#pragma unroll
            for (int i = 0; i < MF; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (TERMS == 1 && i == 0 && reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        }

        if (c == chunks - 1) {
            const tile_pos tp = decode_tile_s2(p, tile_of(q), S2W_ROWS);
            int le = lane;
            asm volatile("" : "+v"(le));
            const int g = le >> 5;
            float* yb = p.y + ((size_t)tp.n * p.m + tp.mt * TM) * plane_out + (size_t)(tp.y0 + wave) * p.w + tp.x0 + (le & 31);
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int m = hf * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                    yb[(size_t)m * plane_out] = TERMS == 4 ? __builtin_ldexpf(acc[hf][e], eu) : acc[hf][e];
                    acc[hf][e] = 0.f;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv3x3_s2_pairs_kernel: the same producer / consumer scheme with a tile of 8 output rows x 32 px x 128 output channels.
//
// Why a second form: the lab's ablations of the kernel above (profiles/r02_conv_s2_lab_ws_ablations.log) show consumers alone at
// 0.51 ms, the producer + DMA pipeline alone at 0.82 ms and both together at 0.95 ms for the 128 -> 64 layer: per 16-channel chunk
// a CU has to pull 37 KiB of x and 36 KiB of weights through its memory path for only 216 MFMAs (42 bytes per MFMA-pipe cycle;
// the stride-1 kernel needs 10.7) and sustains about 20.  The tile has to give each byte more MFMAs, and LDS has to hold it twice:
//   * 128 output channels and 8 output rows quadruple the MFMAs per (x, weight) byte pair;
//   * a chunk is 8 input channels, and the MFMA's k = 16 is (two taps) x (8 channels): lanes 0-31 of the B operand read the pixel of
//     tap 2j, lanes 32-63 the pixel of tap 2j+1 (same LDS word format, different address per lane half), the A operand holds the
//     two taps' weights.  Nine taps = five pairs, the tenth tap has zero weights (10 % of the MFMAs are padding);
//   * LDS image: x [hl][17 rows][2 column parities][33] x 16 B = 35.1 KiB + weights [hl][5 pairs][2][128 m] x 16 B = 40 KiB; two
//     images = 150.1 KiB.  Per chunk 89 KiB of traffic (incl. the idle lanes' loads) for 480 MFMAs: 23 bytes per cycle at full rate.
// A consumer wave owns 2 output rows x 128 m = 8 accumulator tiles; per pair 12 operand reads (8 A + 4 B) against 24 MFMAs, fetched one
// pair ahead.
constexpr int P2_TM = 128;
constexpr int P2_KC = 8;
constexpr int P2_ROWS = 8;
constexpr int P2_RIN = 2 * P2_ROWS + 1;
constexpr int P2_WS_WORDS = 2 * 5 * 2 * P2_TM;            // [hl][pair][tap in pair][128 m]
// S = samples per 32-pixel tile row: 1 (W % 32 == 0: a 32-px segment of one sample) or 2 / 4 (W = 16 / 8: whole rows of S consecutive samples side
// by side).  A parity plane row holds S x (W + 1) = 32 + S words (sample s at s * (W + 1); the odd plane leaves one word per sample unused).
constexpr int p2_pw(int s) { return 32 + s; }
constexpr int P2_EP_WORDS = P2_TM / 4;                      // the tile's 128 bias values (EPI = 1), behind the weights of the tile's last chunk
constexpr int p2_image_words(int s) { return 2 * P2_RIN * 2 * p2_pw(s) + P2_WS_WORDS + P2_EP_WORDS; }
constexpr int p2_lds_bytes(int s) { return 2 * p2_image_words(s) * 16; }
constexpr int P2_LDS_BYTES = p2_lds_bytes(1);

// fp32 [M, K, 3, 3] -> bf16 hi/lo in [m tile of 128][chunk of 8 k][hl][pair][tap in pair][128 m][8 k]; tap 9 is zero.
__global__ __launch_bounds__(256) void conv3x3_prep_weights_pairs(const float* w, u32x4* out, int m_total, int k_total, int terms, const float* w_amax = nullptr) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int chunks = k_total / P2_KC;
    const int total = (m_total / P2_TM) * chunks * 10 * P2_TM;
    if (idx >= total) return;
    int r = idx;
    const int mi = r % P2_TM; r /= P2_TM;
    const int tap = r % 10; r /= 10;
    const int c = r % chunks;
    const int mt = r / chunks;
    const int m = mt * P2_TM + mi, k0 = c * P2_KC;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = tap < 9 ? w[((size_t)m * k_total + k0 + j) * 9 + tap] : 0.f;
    u32x4 hi, lo;
    if (terms == 4) split8t<4>(v, split_scale(amax_exponent(*w_amax)), hi, lo);
    else if (terms == 2) split8t<2>(v, 1.f, hi, lo);      // fp16 tensors: the weights rounded to fp16
    else split8(v, hi, lo);
    const size_t base = ((size_t)mt * chunks + c) * P2_WS_WORDS;
    out[base + (0 * 10 + tap) * P2_TM + mi] = hi;
    if (terms > 2) out[base + (1 * 10 + tap) * P2_TM + mi] = lo;
}

template <int S>
__device__ __forceinline__ tile_pos decode_tile_p2(const s2_params& p, int tile) {   // tp.n = first sample of the tile
    const int mts = p.m / P2_TM, rbs = p.h / P2_ROWS;
    tile_pos tp;
    tp.mt = tile % mts;
    int r = tile / mts;
    if (S == 1) { const int segs = p.w / SEG; tp.x0 = (r % segs) * SEG; r /= segs; } else tp.x0 = 0;
    tp.y0 = (r % rbs) * P2_ROWS;
    tp.n = (r / rbs) * S;
    return tp;
}

// EPI = 1: the layer's tail on the accumulators before the store (layers.py Conv2dLayer.forward + the residual add of DiscriminatorBlock.forward,
// networks.py:343-345):  a = clamp(lrelu_alpha(acc + bias[m]) * gain),  y = a  or, with `accumulate`, y += a  -- a evaluated as max(fma(acc, g, b*g),
// fma(acc, g*alpha, b*g*alpha)) (valid for gain > 0, 0 <= alpha <= 1; alpha = 1 is the linear activation).  The residual add is the reference's own
// in-place `y.add_(x)`: y holds the skip branch's result and every element receives exactly one no-return `global_atomic_add_f32` (fire and forget
// like a store; a read-add-write in the epilogue measured 2.2x the kernel time because the MFMA waves sit on the load latency).  `act_out`
// (optional) receives a, which the backward pass needs once the sum hides it.
struct s2_epilogue {
    const float* bias;       // [m] or NULL
    float* act_out;          // [n, m, h, w] or NULL
    int act;                 // 1 linear, 3 lrelu
    float alpha, gain, clamp;   // clamp < 0: none
    int accumulate;          // y += a instead of y = a
};

// ABL (lab only): 6 consumers only keep the barrier protocol, 7 producers and DMA only keep it, 8 the full kernel without its stores, 10 = 7 + 8.
// IO: element format of x / y / act_out (sgv_io16.h: 0 fp32, 1 bf16, 2 fp16; 16-bit tensors with TERMS = 1 and without `accumulate`) -- same number of
// load instructions (dwordx2 at 2-byte aligned addresses instead of dwordx4 at 4-byte aligned ones), so the counted waits are those of the fp32 form.
template <int TERMS, int ABL = 0, int EPI = 0, int S = 1, int IO = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_s2_pairs_kernel(s2_params p, s2_epilogue ep) {
    constexpr int F = sgv_conv::operand_format<TERMS, IO>();      // operand format of the products (sgv_split.h): TERMS, or 2 = fp16 operands for fp16 tensors
    using namespace sgv_io;
    static_assert(IO == 0 || TERMS == 1, "16-bit tensors are single bf16 operands");
    constexpr int PW = p2_pw(S), P2_XS_PLANE = P2_RIN * 2 * PW, P2_XS_WORDS = 2 * P2_XS_PLANE, P2_IMAGE_WORDS = p2_image_words(S);
    constexpr int SW = 32 / S;          // output pixels per sample in a tile row
    constexpr int GPS = SW / 2;         // 4-column input groups per sample row (without the last, odd column)
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int chunks = p.k / P2_KC;
    const int win = 2 * p.w + 1;
    const size_t plane_in = (size_t)(2 * p.h + 1) * win, plane_out = (size_t)p.h * p.w;
    const int ex = operand_exponent<TERMS>(p.x_amax), ew = operand_exponent<TERMS>(p.w_amax);   // TERMS = 4: block exponents (sgv_split.h)
    const float xS = split_scale(ex);
    const int eu = unscale_exponent(ex, ew);

    const int first = xcd_swizzle(blockIdx.x, gridDim.x);
    if (first >= p.tiles) return;
    const int my_tiles = (p.tiles - first + p.grid - 1) / p.grid;
    const int total = my_tiles * chunks;
    auto tile_of = [&](int q) { return first + (q / chunks) * p.grid; };

    if (wave == 7) {
        // =========================================== weight DMA wave ===========================================
        auto dma_w = [&](int q, u32x4* img) {
            const tile_pos tp = decode_tile_p2<S>(p, tile_of(q));
            const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + (q % chunks)) * P2_WS_WORDS + lane;
            u32x4* wl = img + P2_XS_WORDS;
#pragma unroll
            for (int j = 0; j < P2_WS_WORDS / 64; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wq + j * 64),
                                                 (__attribute__((address_space(3))) void*)(wl + j * 64), 16, 0, 0);
            // EPI = 1: the tile's bias vector rides with its last chunk (read from LDS by the consumers' epilogue: as global loads there they cost the
            // consumer wave a full memory latency per tile -- and, loads and stores sharing vmcnt on gfx9, the drain of the previous tile's stores)
            if (EPI == 1 && ep.bias && (q % chunks) == chunks - 1 && lane < P2_EP_WORDS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const u32x4*)(ep.bias + tp.mt * P2_TM) + lane),
                                                 (__attribute__((address_space(3))) void*)(wl + P2_WS_WORDS), 16, 0, 0);
        };
        if ((ABL != 7 && ABL != 10)) dma_w(0, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // image 0 ready
        for (int q = 0; q < total; q++) {
            if (q + 1 < total && (ABL != 7 && ABL != 10)) dma_w(q + 1, lds + ((q + 1) & 1) * P2_IMAGE_WORDS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    if (wave >= 4) {
        // =========================================== x waves (4, 5, 6) ===========================================
        // 289 units per chunk: 272 items of 8 channels x 4 columns (17 rows x 16 groups) and 17 items for column 64 (the last element of the
        // group that ends there).  Thread i of 192 takes units i and i + 192; every thread issues the same 16 loads per chunk.
        const int pt = t - 256;
        constexpr int ITEMS = 16 * P2_RIN;
        const int u1 = pt + 192;
        const int kind1 = u1 < ITEMS ? 0 : (u1 < ITEMS + S * P2_RIN ? 1 : 2);
        const int a_grp = pt & 15, a_row0 = pt >> 4, a_row1 = u1 >> 4;
        const int a_s = a_grp / GPS, a_g = a_grp % GPS;                      // sample inside the tile row, group inside the sample
        const int h_row = (u1 - ITEMS) % P2_RIN, h_s = kind1 == 1 ? (u1 - ITEMS) / P2_RIN : 0;
        const size_t sample = (size_t)p.k * plane_in;                         // elements between consecutive samples
        struct xset { px4<IO> a[8]; px4<IO> b[8]; };

        auto load_x = [&](int q, xset& r) {
            const tile_pos tp = decode_tile_p2<S>(p, tile_of(q));
            const int c = q % chunks;
            const char* xb_ = at<IO>(p.x, ((size_t)tp.n * p.k + c * P2_KC) * plane_in + (size_t)(2 * tp.y0) * win + 2 * tp.x0);
            const char* q0 = xb_ + (a_s * sample + (size_t)a_row0 * win + 4 * a_g) * fmt<IO>::ES;
            const char* q1 = xb_ + (kind1 == 0 ? a_s * sample + (size_t)a_row1 * win + 4 * a_g : kind1 == 1 ? h_s * sample + (size_t)h_row * win + 2 * SW - 3 : 0) * fmt<IO>::ES;
            const size_t cstep = plane_in * fmt<IO>::ES;
#pragma unroll
            for (int j = 0; j < 8; j++) px4_load<IO>(r.a[j], q0 + j * cstep);
#pragma unroll
            for (int j = 0; j < 8; j++) px4_load<IO>(r.b[j], q1 + j * cstep);
        };
        auto arrive = [&](xset& r, bool newer) {
            if (newer) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) { px4_pin<IO>(r.a[j]); px4_pin<IO>(r.b[j]); }
        };
        auto put = [&](u32x4* xs, int pos, const float* v) {
            u32x4 hi, lo;
            split8t<F>(v, xS, hi, lo);
            xs[pos] = hi;
            if (TERMS > 1) xs[P2_XS_PLANE + pos] = lo;
        };
        auto put_item = [&](u32x4* xs, int row, int grp, const px4<IO>* src) {   // grp: group inside sample a_s
            const int base = row * 2 * PW + a_s * (SW + 1) + 2 * grp;
#pragma unroll
            for (int px = 0; px < 4; px++) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = px4_get<IO>(src[j], px);
                put(xs, base + (px & 1) * PW + (px >> 1), v);
            }
        };
        auto store_x = [&](u32x4* xs, const xset& r) {
            put_item(xs, a_row0, a_g, r.a);
            if (kind1 == 0) put_item(xs, a_row1, a_g, r.b);
            else if (kind1 == 1) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = px4_get<IO>(r.b[j], 3);
                put(xs, h_row * 2 * PW + h_s * (SW + 1) + SW, v);
            }
        };
        auto step = [&](int q, xset& ld, xset& st) {
            const bool more = q + 2 < total && (ABL != 7 && ABL != 10);
            if (more) load_x(q + 2, ld);
            if (q + 1 < total) {
                arrive(st, more);
                if ((ABL != 7 && ABL != 10)) store_x(lds + ((q + 1) & 1) * P2_IMAGE_WORDS, st);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

        xset s0, s1;
        load_x(0, s0);
        if (total > 1 && (ABL != 7 && ABL != 10)) load_x(1, s1);
        arrive(s0, total > 1 && (ABL != 7 && ABL != 10));
        store_x(lds, s0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // image 0 ready
        for (int q = 0; q < total; q += 2) {
            step(q, s0, s1);
            if (q + 1 < total) step(q + 1, s1, s0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =========================================== consumers ===========================================
    f32x16 acc[2][4];   // [row][m quarter]
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int mq = 0; mq < 4; mq++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][mq][e] = 0.f;

    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_s_barrier();   // image 0 ready
    asm volatile("" ::: "memory");
    for (int q = 0; q < total; q++) {
        const u32x4* xs = lds + (q & 1) * P2_IMAGE_WORDS;
        const u32x4* ws = xs + P2_XS_WORDS;
        const int c = q % chunks;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 5;
        const int a_lane = g * P2_TM + (ln & 31);                          // + ((hl * 5 + pair) * 2) * P2_TM + mq * 32
        const int b_lane = (2 * wave) * 4 * PW + ((ln & 31) / SW) * (SW + 1) + (ln & 31) % SW;   // + r * 4 * PW + tap offset of this lane half

        if (ABL != 6) {
        u32x4 a[2][4][2];    // [buffer][m quarter][hl]
        u32x4 b[2][2][2];    // [buffer][row][hl]
        auto tap_off = [](int tap) { const int ky = tap / 3, kx = tap % 3; return (ky * 2 + (kx & 1)) * PW + (kx >> 1); };
        auto fetch = [&](int buf, int pair) {
            const int o0 = tap_off(2 * pair), o1 = tap_off(pair == 4 ? 8 : 2 * pair + 1);   // the padding tap reads tap 8's pixels against zero weights
            const int pos = b_lane + (g ? o1 : o0);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                b[buf][r][0] = xs[pos + r * 4 * PW];
                if (TERMS > 1) b[buf][r][1] = xs[P2_XS_PLANE + pos + r * 4 * PW];
            }
#pragma unroll
            for (int mq = 0; mq < 4; mq++) {
                a[buf][mq][0] = ws[a_lane + ((0 * 5 + pair) * 2) * P2_TM + mq * 32];
                if (TERMS > 1) a[buf][mq][1] = ws[a_lane + ((1 * 5 + pair) * 2) * P2_TM + mq * 32];
            }
        };
        constexpr int RD = TERMS > 1 ? 12 : 6;
        fetch(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
        for (int pair = 0; pair < 5; pair++) {
            const int bb = pair & 1;
            if (pair + 1 < 5) fetch(bb ^ 1, pair + 1);
            if (TERMS > 1) {
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int mq = 0; mq < 4; mq++) acc[r][mq] = ws_mma<0, F>(a[bb][mq][1], b[bb][r][0], acc[r][mq]);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int mq = 0; mq < 4; mq++) acc[r][mq] = ws_mma<0, F>(a[bb][mq][0], b[bb][r][1], acc[r][mq]);
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int mq = 0; mq < 4; mq++) acc[r][mq] = ws_mma<0, F>(a[bb][mq][0], b[bb][r][0], acc[r][mq]);
            constexpr int MF = TERMS > 1 ? 24 : 8;
            const int reads = pair + 1 < 5 ? RD : 0;
#pragma unroll
            for (int i = 0; i < MF; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        }

        if (c == chunks - 1) {
            const tile_pos tp = decode_tile_p2<S>(p, tile_of(q));
            int le = lane;
            asm volatile("" : "+v"(le));
            const int ge = le >> 5;
            const size_t off0 = ((size_t)(tp.n + (le & 31) / SW) * p.m + tp.mt * P2_TM) * plane_out + (size_t)(tp.y0 + 2 * wave) * p.w + tp.x0 + (le & 31) % SW;
            const float al = ep.act == 3 ? ep.alpha : 1.f;
            const float g0 = ep.gain, g1 = ep.gain * al;
            const float g0s = TERMS == 4 ? __builtin_ldexpf(g0, eu) : g0, g1s = TERMS == 4 ? __builtin_ldexpf(g1, eu) : g1;   // ... on accumulators that carry the block scales
            // all sixteen bias vectors of the tile first (from the LDS copy the DMA wave made with this chunk's weights)
            const float* bias_lds = (const float*)(ws + P2_WS_WORDS);
            f32x4 bvs[4][4];
#pragma unroll
            for (int mq = 0; mq < 4; mq++)
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    bvs[mq][e4] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (EPI == 1 && ep.bias) bvs[mq][e4] = *(const f32x4*)(bias_lds + mq * 32 + 8 * e4 + 4 * ge);
                }
            // v_med3_f32 turns a NaN into -clamp: bias_act's own result when there IS a clamp (bias_act.cu:142), wrong without one (the reference
            // propagates the NaN) -- so the un-clamped store loop is its own instantiation behind one wave-uniform branch per tile
            auto store_tile = [&](auto clamped) {
                constexpr bool CLAMP = decltype(clamped)::value;
                const float clamp_hi = ep.clamp;
#pragma unroll
                for (int mq = 0; mq < 4; mq++)
#pragma unroll
                    for (int e4 = 0; e4 < 4; e4++) {
                        const int m0 = mq * 32 + 8 * e4 + 4 * ge;
                        const f32x4 bv = bvs[mq][e4];
#pragma unroll
                        for (int ei = 0; ei < 4; ei++)
#pragma unroll
                            for (int r = 0; r < 2; r++) {
                                const size_t idx = (size_t)(m0 + ei) * plane_out + (size_t)r * p.w;
                                float v = acc[r][mq][4 * e4 + ei];
                                if (TERMS == 4 && EPI != 1) v = __builtin_ldexpf(v, eu);
                                if (EPI == 1) {
                                    v = fmaxf(__builtin_fmaf(v, g0s, bv[ei] * g0), __builtin_fmaf(v, g1s, bv[ei] * g1));
                                    if (CLAMP) v = __builtin_amdgcn_fmed3f(v, -clamp_hi, clamp_hi);
                                    if (ep.act_out) out_store<IO>(ep.act_out, off0 + idx, v);
                                }
                                if (ABL == 8 || ABL == 10) asm volatile("" :: "v"(v));   // lab: no stores
                                else if (IO == 0 && EPI == 1 && ep.accumulate) atomicAdd(p.y + off0 + idx, v); else out_store<IO>(p.y, off0 + idx, v);
                                acc[r][mq][4 * e4 + ei] = 0.f;
                            }
                    }
            };
            if (EPI == 1 && ep.clamp >= 0.f) store_tile(std::true_type{}); else store_tile(std::false_type{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------------------------------
// convT3x3_s2_ws_kernel: producer / consumer form of convT3x3_s2_kernel (conv3x3s2_kernel.h; same arithmetic, same prepared weights).
//
//   y[n,m,2Y+ky,2X+kx] += wgt(m,k,ky,kx) * x[n,k,Y,X]           x: HxW -> y: (2H+1)x(2W+1); the last row / column are left to the caller
//
// Reference: the `conv_transpose2d(stride=2)` in front of the FIR of the up-sampling path (conv2d_resample.py:128-137) and the data gradient of
// the down-sampling path's strided convolution (conv2d_gradfix.py:100-118).
//
// Tile 4 input rows x 32 input px x 64 m; 7 waves:
//   waves 0-3 (consumers): one input row each, the four output parity classes x 64 m = 8 accumulator tiles; per tap 6 MFMAs into the tap's
//                          class, operands fetched two taps ahead (three register buffers);
//   waves 4-5 (x producers): 90 units per chunk (80 items of 8 channels x 4 px + 10 left-halo pixels), one per thread, aligned 16-byte loads;
//   wave 6 (weight DMA).
// The x tile is small here (10 KiB), so LDS holds THREE images of 46.3 KiB: chunk q+2 is filled while chunk q is consumed, a fill has a whole
// extra iteration to land (the DMA wave waits for the previous iteration's 36 pieces, not the ones just issued), and one barrier per chunk
// still orders everything (image (q+2) % 3 was last read in iteration q-1).
constexpr int TW_ROWS = 4;
constexpr int TW_RIN = TW_ROWS + 1;                      // rows y0-1 .. y0+3
// S = samples per 32-pixel tile row: 1 (W % 32 == 0: a 32-px segment of one sample, columns x0-1 .. x0+31) or 2 / 4 (W = 16 / 8: whole rows of
// S consecutive samples side by side, each with its own zero halo word -- the small-resolution layers that went to the vendor library before)
constexpr int tw_pw(int s) { return 32 + s; }
constexpr int tw_image_words(int s) { return 4 * TW_RIN * tw_pw(s) + WS_WORDS; }
constexpr int TW_IMAGES = 3;
constexpr int tw_lds_bytes(int s) { return TW_IMAGES * tw_image_words(s) * 16; }
constexpr int TW_LDS_BYTES = tw_lds_bytes(1);

template <int S>
__device__ __forceinline__ tile_pos decode_tile_tw(const s2_params& p, int tile) {   // tp.n = first sample of the tile
    const int mts = (p.m + TM - 1) / TM, rbs = p.h / TW_ROWS;     // (a last m tile may be half full: m % 32 == 0)
    tile_pos tp;
    tp.mt = tile % mts;
    int r = tile / mts;
    if (S == 1) { const int segs = p.w / SEG; tp.x0 = (r % segs) * SEG; r /= segs; } else tp.x0 = 0;
    tp.y0 = (r % rbs) * TW_ROWS;
    tp.n = (r / rbs) * S;
    return tp;
}

// ABL (lab only): 6 consumers only keep the barrier protocol, 7 producers and DMA only keep it, 8 the full kernel without its stores, 10 = 7 + 8
// (profiles/r03_s2_lab_store_burst.log: the tile-end stores cost the 64 <- 128 channel layer 21 %, and pairing the even / odd columns into one 8-byte store
// or starting the workgroups staggered changes nothing -- the burst is HBM-write bound, 128 KiB per CU and tile at ~11 B/clk/CU).
// IO: element format of x / y (sgv_io16.h; 16-bit tensors with TERMS = 1): 8-byte aligned dwordx2 loads, two 2-byte stores per accumulator pair.
// CM (tools/convT_lab.hip; the product instantiates 0): which accumulator tiles a consumer wave owns.
//   0: ONE input row x 4 parity classes x both 32-channel halves of the 64 output channels.  Per tap 4 weight reads + 2 pixel reads feed 6 MFMAs
//      (44 ds_read_b128 per 54 MFMAs and chunk).
//   1 (round 5): TWO input rows x 4 classes x ONE half (wave = row pair * 2 + half); the six pixel operands of the pair (input rows r0-1 .. r0+1, columns
//      x and x-1) stay in registers for the chunk and a tap's weights are read once for both rows: 30 reads per 54 MFMAs, the stride-1 kernel's ratio.
//   2: as 1, and the first two pixel operands of the next chunk are read before the barrier (the x part of image q + 1 is complete by then).
//   Measured (profiles/r05_c1_convT_lab.log, fp16 split): the consumer loop alone gains 4-5 % from 1 (0.522 vs 0.549 ms on 512 -> 256 channels) and
//   nothing more from 2; the WHOLE kernel is 3-9 % slower with 1 and within +-3 % with 2 (0.873 / 0.753 / 0.707 ms against 0.901 / 0.751 / 0.683): the
//   operand reads are not what the loop waits for.  The same log splits the kernel's time: consumers alone 0.55-0.68 ms, producers + DMA + stores alone
//   0.41-0.86, everything but the stores 0.62-0.64, the tile-end stores 0.06 (256 channels out) - 0.26 ms (64 channels out, 1.6 GB).
template <int TERMS, int ABL = 0, int S = 1, int IO = 0, int CM = 0, int ST = 0>
__global__ __launch_bounds__(448, 2) void convT3x3_s2_ws_kernel(s2_params p) {
    static_assert(ST == 0 || CM == 0, "the store schedule is written for the one-row mapping");
    constexpr int F = sgv_conv::operand_format<TERMS, IO>();      // operand format of the products (sgv_split.h): TERMS, or 2 = fp16 operands for fp16 tensors
    using namespace sgv_io;
    static_assert(IO == 0 || TERMS == 1, "16-bit tensors are single bf16 operands");
    constexpr int TW_PW = tw_pw(S), TW_XS_PLANE = TW_RIN * TW_PW, TW_XS_WORDS = 4 * TW_XS_PLANE, TW_IMAGE_WORDS = tw_image_words(S);
    constexpr int SW = 32 / S;   // pixels per sample in a tile row
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int chunks = p.k / KC;
    const int wout = 2 * p.w + 1;
    const size_t plane_in = (size_t)p.h * p.w, plane_out = (size_t)(2 * p.h + 1) * wout;
    const int ex = operand_exponent<TERMS>(p.x_amax), ew = operand_exponent<TERMS>(p.w_amax);   // TERMS = 4: block exponents (sgv_split.h)
    const float xS = split_scale(ex);
    const int eu = unscale_exponent(ex, ew);

    const int first = xcd_swizzle(blockIdx.x, gridDim.x);
    if (first >= p.tiles) return;
    const int my_tiles = (p.tiles - first + p.grid - 1) / p.grid;
    const int total = my_tiles * chunks;
    auto tile_of = [&](int q) { return first + (q / chunks) * p.grid; };
    auto image = [&](int q) { return lds + (q % TW_IMAGES) * TW_IMAGE_WORDS; };

    if (wave == 6) {
        // =========================================== weight DMA wave ===========================================
        auto dma_w = [&](int q) {
            const tile_pos tp = decode_tile_tw<S>(p, tile_of(q));
            const u32x4* wq = p.wprep + ((size_t)tp.mt * chunks + (q % chunks)) * WS_WORDS + lane;
            u32x4* wl = image(q) + TW_XS_WORDS;
#pragma unroll
            for (int j = 0; j < 36; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wq + j * 64),
                                                 (__attribute__((address_space(3))) void*)(wl + j * 64), 16, 0, 0);
        };
        if ((ABL != 7 && ABL != 10)) { dma_w(0); if (total > 1) dma_w(1); }
        if (total > 1 && (ABL != 7 && ABL != 10)) asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // image 0 ready
        for (int q = 0; q < total; q++) {
            const bool more = q + 2 < total && (ABL != 7 && ABL != 10);
            if (more) dma_w(q + 2);
            // chunk q+1 (issued one iteration ago) must have landed before this barrier; the 36 pieces just issued may stay in flight
            if (more) asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    if (wave >= 4) {
        // =========================================== x waves (4, 5) ===========================================
        const int u = t - 256;                                   // 0..127; units 0..79 items (octet, row, quad), then 10 * S halo words (octet, row, sample)
        const int kind = u < 16 * TW_RIN ? 0 : (u < (16 + 2 * S) * TW_RIN ? 1 : 2);
        const int oct = u & 1;
        const int hh = (u - 16 * TW_RIN) >> 1;
        const int a_quad = (u >> 1) & 7, a_row = kind == 0 ? u >> 4 : kind == 1 ? hh % TW_RIN : 0;
        const int a_s = kind == 0 ? a_quad / (SW / 4) : kind == 1 ? hh / TW_RIN : 0;     // sample inside the tile row
        const int a_q = a_quad % (SW / 4);
        struct xset { px4<IO> a[8]; bool ok; };

        auto load_x = [&](int q, xset& r) {
            const tile_pos tp = decode_tile_tw<S>(p, tile_of(q));
            const int c = q % chunks;
            const int gy = tp.y0 - 1 + a_row;
            const int gx = kind == 0 ? tp.x0 + 4 * a_q : tp.x0 - 4;         // the halo pixel x0-1 is the last element of the group before the tile
            r.ok = kind != 2 && gy >= 0 && gx >= 0 && (S == 1 || kind == 0);   // packed samples: every halo word is image padding
            const char* q0 = at<IO>(p.x, ((size_t)(tp.n + a_s) * p.k + c * KC + 8 * oct) * plane_in + (size_t)max(gy, 0) * p.w + max(gx, 0));
            const size_t cstep = plane_in * fmt<IO>::ES;
#pragma unroll
            for (int j = 0; j < 8; j++) px4_load<IO>(r.a[j], q0 + j * cstep);
        };
        auto arrive = [&](xset& r, bool newer) {
            if (newer) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) px4_pin<IO>(r.a[j]);
        };
        auto put = [&](u32x4* xs, int pos, float* v, bool ok) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = ok ? v[j] : 0.f;
            u32x4 hi, lo;
            split8t<F>(v, xS, hi, lo);
            xs[pos] = hi;
            if (TERMS > 1) xs[2 * TW_XS_PLANE + pos] = lo;
        };
        auto store_x = [&](u32x4* xs, const xset& r) {
            if (kind == 0) {
                const int base = (oct * TW_RIN + a_row) * TW_PW + a_s * (SW + 1) + 1 + 4 * a_q;
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = px4_get<IO>(r.a[j], px);
                    put(xs, base + px, v, r.ok);
                }
            } else if (kind == 1) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = px4_get<IO>(r.a[j], 3);
                put(xs, (oct * TW_RIN + a_row) * TW_PW + a_s * (SW + 1), v, r.ok);
            }
        };
        // iteration q: start the loads of chunk q+3, write chunk q+2 (loaded an iteration ago) into image (q+2) % 3
        auto step = [&](int q, xset& ld, xset& st) {
            const bool more = q + 3 < total && (ABL != 7 && ABL != 10);
            if (more) load_x(q + 3, ld);
            if (q + 2 < total) {
                arrive(st, more);
                if ((ABL != 7 && ABL != 10)) store_x(image(q + 2), st);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

        xset s0, s1;
        load_x(0, s0);
        if (total > 1) load_x(1, s1);
        arrive(s0, total > 1);
        store_x(image(0), s0);
        if (total > 2 && (ABL != 7 && ABL != 10)) load_x(2, s0);
        if (total > 1) { arrive(s1, total > 2 && (ABL != 7 && ABL != 10)); store_x(image(1), s1); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // images 0 and 1 ready
        // chunk q+2 sits in s0 for even q, s1 for odd q
        for (int q = 0; q < total; q += 2) {
            step(q, s1, s0);
            if (q + 1 < total) step(q + 1, s0, s1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =========================================== consumers ===========================================
    if constexpr (CM >= 1) {
    f32x16 acc[2][4];   // [row of the pair][class a*2+b]
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int cl = 0; cl < 4; cl++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][cl][e] = 0.f;
    const int rp = wave >> 1, hf = wave & 1;      // wave-uniform: input rows 2 rp, 2 rp + 1 of the tile; output channels 32 hf .. 32 hf + 31 of its 64
    u32x4 B[3][2][2];    // [jj][dx][hl]: pixel operands of LDS rows 2 rp + jj at columns x - dx
    u32x4 A[3][2];       // [buffer][hl]: the tap's weights for this wave's half, fetched two taps ahead

    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_s_barrier();   // images 0 and 1 ready
    asm volatile("" ::: "memory");
    if (CM == 2 && ABL != 6) {
        const int b0 = ((lane >> 5) * TW_RIN + 2 * rp) * TW_PW + ((lane & 31) / SW) * (SW + 1) + (lane & 31) % SW + 1;
#pragma unroll
        for (int jj = 1; jj <= 2; jj++) {
            B[jj][0][0] = lds[b0 + jj * TW_PW];
            if (TERMS > 1) B[jj][0][1] = lds[2 * TW_XS_PLANE + b0 + jj * TW_PW];
        }
    }
    for (int q = 0; q < total; q++) {
        const u32x4* xs = image(q);
        const u32x4* ws = xs + TW_XS_WORDS;
        const int c = q % chunks;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int a_lane = (ln >> 5) * TM + hf * 32 + (ln & 31);                                                    // + ((hl * 9 + tap) * 2) * TM
        const int b_lane = ((ln >> 5) * TW_RIN + 2 * rp) * TW_PW + ((ln & 31) / SW) * (SW + 1) + (ln & 31) % SW + 1;   // + jj * PW - dx: LDS row 2 rp + jj = tile row 2 rp - 1 + jj

        if (ABL != 6) {
        auto fetch_b = [&](const u32x4* img, int jj, int dx) {
            const int pos = b_lane + jj * TW_PW - dx;
            B[jj][dx][0] = img[pos];
            if (TERMS > 1) B[jj][dx][1] = img[2 * TW_XS_PLANE + pos];
        };
        auto fetch_a = [&](int buf, int tap) {
            if (TERMS > 1) A[buf][1] = ws[a_lane + ((1 * 9 + tap) * 2) * TM];
            A[buf][0] = ws[a_lane + ((0 * 9 + tap) * 2) * TM];
        };
        constexpr int RD = TERMS > 1 ? 2 : 1;   // reads per operand
        // in the order of first use: tap 0 (ky = 0, kx = 0) needs rows jj = 1, 2 at dx = 0 and its weights
        if (CM == 1) { fetch_b(xs, 1, 0); fetch_b(xs, 2, 0); }     // CM = 2: fetched behind the previous chunk's last taps (below; chunk 0: in front of the loop)
        fetch_a(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (CM == 1 ? 3 : 1) * RD, 0);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
            const int cl = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
            const int dx = kx == 2 ? 1 : 0;
            const int j0 = ky == 2 ? 0 : 1;          // row r of the pair reads LDS row jj = j0 + r
            const int bb = tap % 3;
            int reads = 0;
            // the weights of the next tap but one, and the remaining pixel operands a tap or more ahead of their first use: (1,1), (2,1) for tap 2;
            // (0,0) for tap 6, (0,1) for tap 8
            if (tap == 0) { fetch_a(1, 1); reads += RD; }
            if (tap == 0) { fetch_b(xs, 1, 1); reads += RD; }
            if (tap == 1) { fetch_b(xs, 2, 1); reads += RD; }
            if (tap == 2) { fetch_b(xs, 0, 0); reads += RD; }
            if (tap == 3) { fetch_b(xs, 0, 1); reads += RD; }
            if (tap + 2 < 9) { fetch_a((tap + 2) % 3, tap + 2); reads += RD; }
            if (CM == 2) {
                // The x part of image q + 1 has been complete since the previous barrier (three images: the producers are filling q + 2), so the first two pixel
                // operands of the NEXT chunk are read behind this chunk's last taps, into registers whose last use has passed -- (2,0) after tap 5, (1,0)
                // after tap 7 -- and the first MFMA behind the barrier waits for its weights only (the DMA wave guarantees those at the barrier, not earlier).
                // Behind the last chunk this reads an image nobody filled: allocated LDS, values never used.
                if (tap == 6) { fetch_b(image(q + 1), 2, 0); reads += RD; }
                if (tap == 8) { fetch_b(image(q + 1), 1, 0); reads += RD; }
            }
            if (TERMS > 1) {
#pragma unroll
                for (int r = 0; r < 2; r++) acc[r][cl] = ws_mma<0, F>(A[bb][1], B[j0 + r][dx][0], acc[r][cl]);
#pragma unroll
                for (int r = 0; r < 2; r++) acc[r][cl] = ws_mma<0, F>(A[bb][0], B[j0 + r][dx][1], acc[r][cl]);
            }
#pragma unroll
            for (int r = 0; r < 2; r++) acc[r][cl] = ws_mma<0, F>(A[bb][0], B[j0 + r][dx][0], acc[r][cl]);
            constexpr int MF = TERMS > 1 ? 6 : 2;
#pragma unroll
            for (int i = 0; i < MF; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        }

        if (c == chunks - 1) {
            const tile_pos tp = decode_tile_tw<S>(p, tile_of(q));
            int le = lane;
            asm volatile("" : "+v"(le));
            const int g = le >> 5;
            const size_t yb = ((size_t)(tp.n + (le & 31) / SW) * p.m + tp.mt * TM + hf * 32) * plane_out + (size_t)(2 * (tp.y0 + 2 * rp)) * wout + 2 * (tp.x0 + (le & 31) % SW);
            const bool live = hf * 32 < p.m - tp.mt * TM;     // (wave-uniform) the upper half of a half-full last m tile has zero weights and is not stored
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int m = (e & 3) + 8 * (e >> 2) + 4 * g;
                        const size_t qd = yb + (size_t)m * plane_out + (size_t)(2 * r + a2) * wout;
                        if (ABL == 8 || ABL == 10) { asm volatile("" :: "v"(acc[r][a2 * 2 + 0][e])); asm volatile("" :: "v"(acc[r][a2 * 2 + 1][e])); }   // lab: no stores
                        else if (live) {
                            out_store<IO>(p.y, qd, TERMS == 4 ? __builtin_ldexpf(acc[r][a2 * 2 + 0][e], eu) : acc[r][a2 * 2 + 0][e]);
                            out_store<IO>(p.y, qd + 1, TERMS == 4 ? __builtin_ldexpf(acc[r][a2 * 2 + 1][e], eu) : acc[r][a2 * 2 + 1][e]);
                        }
                        acc[r][a2 * 2 + 0][e] = 0.f;
                        acc[r][a2 * 2 + 1][e] = 0.f;
                    }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    return;
    }
    if constexpr (ST == 1) {
    // ST = 1 (round 5; tools/convT_lab.hip only -- the product instantiates ST = 0).  Measured (profiles/r05_c6_convT_lab.log, same box): NOT faster -- whole kernel
    // 0.996 / 0.761 / 0.685 ms against 0.955 / 0.753 / 0.688 for the burst form, in the step 14.9 against 14.2 ms per iteration -- while the no-store ablation of
    // either form runs 0.61 - 0.65 ms: what the stores cost is the bytes through the CU's memory pipe (128 KiB per tile, shared with the producers' loads and the
    // weight DMA), wherever the instructions sit in the wave's stream.
    // The tile's 128 stores per wave are folded INTO the MFMA stream instead of being issued as one burst behind the last chunk, where every
    // CU of the chip stores at once at HBM-write speed while its matrix pipe idles (profiles/r05_c1_convT_lab.log: 0.06 - 0.26 ms of 0.68 - 0.90 ms per layer).
    // No extra registers: the nine taps of EVERY chunk run class by class -- oo (tap 4), oe (3, 5), eo (1, 7), ee (0, 2, 6, 8) -- so in a tile's last chunk a class is
    // final as soon as its own taps are done, and its 32 stores (2 halves x 16 accumulator registers) leave between the MFMAs of the classes behind it; the class
    // that finishes last, ee, is stored between the first taps of the NEXT chunk (the next tile's first), whose own ee taps come last.  Same products; the
    // summation order within a chunk is the class order instead of the tap order.
    f32x16 acc[4][2];   // [class a*2+b][m half]
#pragma unroll
    for (int cl = 0; cl < 4; cl++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[cl][hf][e] = 0.f;
    // output addresses = a wave-uniform element offset (tile and accumulator register: scalar registers) + ONE per-lane offset that never changes: a store costs
    // no address arithmetic in vector registers (24 stores in flight with 64-bit vector addresses each would not fit beside 128 accumulators + 72 operand registers)
    const unsigned lane_off = (unsigned)(((size_t)((lane & 31) / SW) * p.m + 4 * (lane >> 5)) * plane_out) + 2u * ((lane & 31) % SW);     // (< 2^31 elements: host-checked tensor size)
    size_t tb_prev = 0;           // the previous tile's uniform offset while its ee class is still to be stored

    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_s_barrier();   // images 0 and 1 ready
    asm volatile("" ::: "memory");
    // stores [u0, u1) of class cl (unit u: half u >> 4, accumulator register u & 15) of the tile at uniform offset tb; the registers are cleared as they leave.
    // (Whole 64-channel tiles only: the host keeps half-full last tiles on the burst form, ST = 0.)
    auto store_units = [&](int cl, int u0, int u1, size_t tb) {
#pragma unroll
        for (int u = u0; u < u1; u++) {
            const int hf = u >> 4, e = u & 15;
            const size_t ub = tb + (size_t)(hf * 32 + (e & 3) + 8 * (e >> 2)) * plane_out + (size_t)(cl >> 1) * wout + (cl & 1);      // wave-uniform
            if (ABL == 8 || ABL == 10) asm volatile("" :: "v"(acc[cl][hf][e]));   // lab: no stores
            else out_store_lane<IO>((char*)p.y + ub * fmt<IO>::ES, lane_off, acc[cl][hf][e]);
            acc[cl][hf][e] = 0.f;
        }
    };
    // a class is final: undo the operands' block scales IN PLACE (TERMS = 4), so that the stores behind read the accumulator registers themselves -- a scaled copy
    // per store in flight is 12 - 24 more live registers than the kernel has
    auto finalize = [&](int cl) {
        if (TERMS != 4) return;
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[cl][hf][e] = __builtin_ldexpf(acc[cl][hf][e], eu);
    };
    // One chunk.  LAST: the tile's last chunk (stores of oo / oe / eo behind their last taps); PEND: the previous tile's ee class is stored behind the first taps.
    // Compile-time flags, and the three kinds of chunk of a tile -- first (PEND), middle, last (LAST) -- are three consecutive pieces of code in the tile loop,
    // not the arms of a branch: the compiler hoists the common MFMA / read sequence out of an if / else over whole chunks and leaves the stores behind in
    // conditional blocks, which loses the order below and spills 200 registers.
    auto chunk = [&](int q, size_t tb, auto last_c, auto pend_c) {
        constexpr bool LAST = decltype(last_c)::value, PEND = decltype(pend_c)::value;
        const u32x4* xs = image(q);
        const u32x4* ws = xs + TW_XS_WORDS;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int a_lane = (ln >> 5) * TM + (ln & 31);
        const int b_lane = ((ln >> 5) * TW_RIN + wave + 1) * TW_PW + ((ln & 31) / SW) * (SW + 1) + (ln & 31) % SW + 1;      // - dy * PW - dx
        if (ABL != 6) {
            constexpr int ORD[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};      // oo | oe oe | eo eo | ee ee ee ee
            u32x4 a[3][2][2];    // [buffer][half][hl]
            u32x4 b[3][2];       // [buffer][hl]
            auto fetch = [&](int buf, int tap) {
                const int ky = tap / 3, kx = tap % 3;
                const int pos = b_lane - (ky == 2 ? TW_PW : 0) - (kx == 2 ? 1 : 0);
                b[buf][0] = xs[pos];
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    if (TERMS > 1) a[buf][hf][1] = ws[a_lane + ((1 * 9 + tap) * 2) * TM + hf * 32];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) a[buf][hf][0] = ws[a_lane + ((0 * 9 + tap) * 2) * TM + hf * 32];
                if (TERMS > 1) b[buf][1] = xs[2 * TW_XS_PLANE + pos];
            };
            constexpr int RD = TERMS > 1 ? 6 : 3;
            fetch(0, ORD[0]);
            fetch(1, ORD[1]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * RD, 0);
#pragma unroll
            for (int ps = 0; ps < 9; ps++) {
                const int tap = ORD[ps];
                const int ky = tap / 3, kx = tap % 3;
                const int cl = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
                const int bb = ps % 3;
                if (ps + 2 < 9) fetch((ps + 2) % 3, ORD[ps + 2]);
                // stores that leave behind this position's MFMAs: the previous tile's ee class (class 0) in positions 0 - 2; this tile's oo (class 3), final after
                // position 0, in 1 - 3; oe (class 2), final after position 2, in 3 - 6; eo (class 1), final after position 4, in 6 - 8
                int stores = 0;
                if (PEND) {
                    if (ps == 0) { store_units(0, 0, 12, tb_prev); stores += 12; }
                    if (ps == 1) { store_units(0, 12, 24, tb_prev); stores += 12; }
                    if (ps == 2) { store_units(0, 24, 32, tb_prev); stores += 8; }
                }
                if (LAST) {
                    if (ps == 1) finalize(3);
                    if (ps == 3) finalize(2);
                    if (ps == 5) finalize(1);
                    if (ps == 1) { store_units(3, 0, 12, tb); stores += 12; }
                    if (ps == 2) { store_units(3, 12, 24, tb); stores += 12; }
                    if (ps == 3) { store_units(3, 24, 32, tb); store_units(2, 0, 4, tb); stores += 12; }
                    if (ps == 4) { store_units(2, 4, 16, tb); stores += 12; }
                    if (ps == 5) { store_units(2, 16, 28, tb); stores += 12; }
                    if (ps == 6) { store_units(2, 28, 32, tb); store_units(1, 0, 8, tb); stores += 12; }
                    if (ps == 7) { store_units(1, 8, 20, tb); stores += 12; }
                    if (ps == 8) { store_units(1, 20, 32, tb); stores += 12; }
                }
                if (TERMS > 1) {
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) acc[cl][hf] = ws_mma<0, F>(a[bb][hf][1], b[bb][0], acc[cl][hf]);
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) acc[cl][hf] = ws_mma<0, F>(a[bb][hf][0], b[bb][1], acc[cl][hf]);
                }
#pragma unroll
                for (int hf = 0; hf < 2; hf++) acc[cl][hf] = ws_mma<0, F>(a[bb][hf][0], b[bb][0], acc[cl][hf]);
                constexpr int MF = TERMS > 1 ? 6 : 2;
                const int reads = ps + 2 < 9 ? RD : 0;
                const int per_gap = (stores + MF - 1) / MF;
#pragma unroll
                for (int i = 0; i < MF; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (TERMS == 1 && i == 0 && reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (ABL != 8 && ABL != 10) {      // up to six stores behind this MFMA (the group size must be a literal)
#pragma unroll
                        for (int k = 0; k < 6; k++)
                            if (k < per_gap && i * per_gap + k < stores) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                    }
                }
            }
            if (LAST) finalize(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // tiles of this workgroup; chunks >= 2 (host-checked).  The first tile has no predecessor: its "previous tile" is itself -- the zeros that leave for its ee
    // outputs are overwritten by the same wave's real stores later (stores of one wave to one address keep their order).
    int q = 0;
    for (int tl = 0; tl < my_tiles; tl++) {
        const tile_pos tp = decode_tile_tw<S>(p, tile_of(q));
        const size_t tb = ((size_t)tp.n * p.m + tp.mt * TM) * plane_out + (size_t)(2 * (tp.y0 + wave)) * wout + 2 * tp.x0;     // wave-uniform
        if (tl == 0) tb_prev = tb;
        chunk(q++, tb, std::false_type{}, std::true_type{});
        for (int c = 1; c < chunks - 1; c++) chunk(q++, tb, std::false_type{}, std::false_type{});
        chunk(q++, tb, std::true_type{}, std::false_type{});
        tb_prev = tb;
    }
    store_units(0, 0, 32, tb_prev);     // the last tile's ee class
    return;
    }
    f32x16 acc[4][2];   // [class a*2+b][m half]
#pragma unroll
    for (int cl = 0; cl < 4; cl++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[cl][hf][e] = 0.f;

    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_s_barrier();   // images 0 and 1 ready
    asm volatile("" ::: "memory");
    for (int q = 0; q < total; q++) {
        const u32x4* xs = image(q);
        const u32x4* ws = xs + TW_XS_WORDS;
        const int c = q % chunks;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int a_lane = (ln >> 5) * TM + (ln & 31);
        const int b_lane = ((ln >> 5) * TW_RIN + wave + 1) * TW_PW + ((ln & 31) / SW) * (SW + 1) + (ln & 31) % SW + 1;      // - dy * PW - dx

        if (ABL != 6) {
        u32x4 a[3][2][2];    // [buffer][half][hl]
        u32x4 b[3][2];       // [buffer][hl]
        auto fetch = [&](int buf, int tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int pos = b_lane - (ky == 2 ? TW_PW : 0) - (kx == 2 ? 1 : 0);   // tap 2 reaches back to the previous input row / column
            b[buf][0] = xs[pos];
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
                if (TERMS > 1) a[buf][hf][1] = ws[a_lane + ((1 * 9 + tap) * 2) * TM + hf * 32];
#pragma unroll
            for (int hf = 0; hf < 2; hf++) a[buf][hf][0] = ws[a_lane + ((0 * 9 + tap) * 2) * TM + hf * 32];
            if (TERMS > 1) b[buf][1] = xs[2 * TW_XS_PLANE + pos];
        };
        constexpr int RD = TERMS > 1 ? 6 : 3;
        fetch(0, 0);
        fetch(1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * RD, 0);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
            const int cl = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
            const int bb = tap % 3;
            if (tap + 2 < 9) fetch((tap + 2) % 3, tap + 2);
            if (TERMS > 1) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++) acc[cl][hf] = ws_mma<0, F>(a[bb][hf][1], b[bb][0], acc[cl][hf]);
#pragma unroll
                for (int hf = 0; hf < 2; hf++) acc[cl][hf] = ws_mma<0, F>(a[bb][hf][0], b[bb][1], acc[cl][hf]);
            }
#pragma unroll
            for (int hf = 0; hf < 2; hf++) acc[cl][hf] = ws_mma<0, F>(a[bb][hf][0], b[bb][0], acc[cl][hf]);
            constexpr int MF = TERMS > 1 ? 6 : 2;
            const int reads = tap + 2 < 9 ? RD : 0;
#pragma unroll
            for (int i = 0; i < MF; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (TERMS == 1 && i == 0 && reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        }

        if (c == chunks - 1) {
            const tile_pos tp = decode_tile_tw<S>(p, tile_of(q));
            int le = lane;
            asm volatile("" : "+v"(le));
            const int g = le >> 5;
            const size_t yb = ((size_t)(tp.n + (le & 31) / SW) * p.m + tp.mt * TM) * plane_out + (size_t)(2 * (tp.y0 + wave)) * wout + 2 * (tp.x0 + (le & 31) % SW);
            const int m_left = p.m - tp.mt * TM;     // 64, or 32 in a half-full last m tile (wave-uniform): the padded half has zero weights and is not stored
#pragma unroll
            for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int m = hf * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                        const size_t qd = yb + (size_t)m * plane_out + (size_t)a2 * wout;
                        if (hf * 32 >= m_left) { acc[a2 * 2 + 0][hf][e] = 0.f; acc[a2 * 2 + 1][hf][e] = 0.f; continue; }
                        if (ABL == 8 || ABL == 10) { asm volatile("" :: "v"(acc[a2 * 2 + 0][hf][e])); asm volatile("" :: "v"(acc[a2 * 2 + 1][hf][e])); }   // lab: no stores
                        else {
                            out_store<IO>(p.y, qd, TERMS == 4 ? __builtin_ldexpf(acc[a2 * 2 + 0][hf][e], eu) : acc[a2 * 2 + 0][hf][e]);
                            out_store<IO>(p.y, qd + 1, TERMS == 4 ? __builtin_ldexpf(acc[a2 * 2 + 1][hf][e], eu) : acc[a2 * 2 + 1][hf][e]);
                        }
                        acc[a2 * 2 + 0][hf][e] = 0.f;
                        acc[a2 * 2 + 1][hf][e] = 0.f;
                    }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

}  // namespace sgv_conv
