// Weight gradient of a 3x3 / stride 1 / pad 1 convolution -- producer / consumer form of wrw3x3_kernel (wrw_kernel.h), same arithmetic
// (bf16 hi/lo split products on v_mfma_f32_32x32x16_bf16, fp32 accumulate), same work decomposition (64 o x 64 i x 9 taps per workgroup,
// persistent over (sample, 32-pixel column segment, row block) units, one atomicAdd per element at the end).
//
//     dw[o,i,ky,kx] = sum_{n,y,x} dy[n,o,y,x] * x[n,i,y+ky-1,x+kx-1]
//
// Reference: `Conv2dGradWeight` of conv2d_gradfix.py:140-170 for the stride-1 layers of networks.py / layers.py.
//
// Why: the 4-wave kernel spends 4.0k cycles per row step on 1.7k cycles of MFMA issue -- every wave loads, splits, writes LDS, builds the
// kx = 0 / 2 operand views with v_alignbyte and only then feeds the matrix pipe (one wave per SIMD, in order).  Same cure as
// conv3x3_ws_kernel.h: 8 waves,
//   waves 0-3 (consumers, 2 x 2 over the 64 x 64 tile): aligned 16-B LDS operand reads + MFMAs only (144 accumulators); operands are
//                   fetched one (k half, ky) sub-step ahead, the first sub-step of row y+1 is fetched before the barrier that ends row y;
//   waves 4-7 (producers): global loads (inline asm, counted vmcnt, two register sets: a row is in flight for a whole step), hi/lo split,
//                   and the THREE column-shifted views of the x row (kx = 0, 1, 2) written as aligned 16-B words, so that the consumers
//                   need no VALU at all.
// LDS: x ring 4 rows x {hi,lo} x 3 views x 64 ch x 80 B = 120 KiB (rows y-1, y, y+1 in use, y+2 being written), dy 3 rows x {hi,lo}
// x 64 ch x 80 B = 30 KiB (dy runs two rows ahead so that the next step's first operands are readable before the barrier): 150 KiB.
#pragma once

#include "wrw_kernel.h"
#include "sgv_io16.h"

#ifndef SGV_WRW_RPM
#define SGV_WRW_RPM 2
#endif

namespace sgv_wrw {

// VIEWS = 3: the producers write three column-shifted copies of every x row (kx = 0, 1, 2) and the consumers issue aligned 16-B reads only;
// VIEWS = 1: one copy (plus the two halo pixels), the consumers build the kx = 0 / 2 operands with v_alignbyte_b32 as the 4-wave kernel does:
//            less than half the LDS traffic per row step (88 vs 192 KiB), ten VALU operations per nine MFMAs in the consumer waves.
constexpr int WS_VIEW = TI * RS + 8;             // bf16 per (slot, hl, view); the right halo of a row sits in the first (unused) word of the next row
constexpr int WS_DBUF = TO * RS;                 // bf16 per (buffer, hl)
constexpr int WS_DS = 2 * 3 * WS_DBUF;           // [hl][3 buffers][64][RS]
constexpr int wrw_ws_lds_bytes(int views) { return (2 * 4 * views * WS_VIEW + WS_DS) * 2; }
constexpr int WRW_WS_LDS_BYTES = wrw_ws_lds_bytes(3);

// ABL (tools/wrw_lab.hip only; wrong results by construction): 6 consumers only keep the barrier protocol, 7 producers only keep it, 8 no final flush,
// 10 producers without the split arithmetic, 11 producers without global loads; 12 = 6 + 10; 13 = 6 with opaque register values instead of loads, 14 = the full kernel with them.
//
// PACK: images 16 or 8 pixels wide (the < 32^2 layers): 2 or 4 samples sit side by side in the 32-pixel row step, a unit is (group of 32 / W
// samples, row block).  The producers take every 8-pixel group from its own sample (left / right neighbours outside the sample's row are zero),
// the consumers clear the one pixel that the kx = 0 / 2 shifts would otherwise pull across a sample boundary (VIEWS = 1 only).
// IO: element format of dy and x (sgv_io16.h): 0 fp32; 1 bf16 / 2 fp16 (TERMS = 1: single bf16 operands; dw and the input scale stay fp32).  Same six
//     loads per item -- dwordx4 -> dwordx2, halo dword -> ushort -- so every counted wait below holds for all formats.
template <int TERMS, int VIEWS = 1, int ABL = 0, bool PACK = false, int IO = 0>
__global__ __launch_bounds__(512, 2) void wrw3x3_ws_kernel(wrw_params p) {
    constexpr int F = sgv_conv::operand_format<TERMS, IO>();      // operand format of the products (sgv_split.h): TERMS, or 2 = fp16 operands for fp16 tensors
    static_assert(!PACK || VIEWS == 1, "packed samples: single-view form only");
    static_assert(IO == 0 || TERMS == 1, "16-bit tensors are multiplied as single bf16 operands");
    using namespace sgv_io;
    constexpr int WS_XSLOT = VIEWS * WS_VIEW;        // bf16 per (slot, hl)
    constexpr int WS_XS = 2 * 4 * WS_XSLOT;          // [hl][4 slots][views][64][RS]
    constexpr int XO = VIEWS == 1 ? XROW0 : 0;       // position of the segment's first pixel inside a row
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_ws[];
    unsigned short* xs = lds_ws;             // ((hl * 4 + slot) * VIEWS + view) * WS_VIEW + ch * RS + XO + px   VIEWS = 3: holds x[col = x0 + px + view - 1]
    unsigned short* ds = lds_ws + WS_XS;     // ((hl * 3 + buf) * 64 + o) * RS + px

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    // XCD-aware: workgroups that walk the same units (same split) on different output tiles share one L2
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int vid = (nwg & 7) == 0 ? (lin & 7) * (nwg >> 3) + (lin >> 3) : lin;
    const int tile = vid % (int)gridDim.x, split = vid / (int)gridDim.x;
    const int o0 = (tile / p.tiles_i) * TO, i0 = (tile % p.tiles_i) * TI;
    const int segs = PACK ? 1 : p.w / SEG, rblocks = p.h / p.rows;
    const int spr = PACK ? SEG / p.w : 1;     // samples per row step
    const size_t plane = (size_t)p.h * p.w;
    const int R = p.rows;
    // TERMS = 4: block exponents of the two operands (sgv_split.h); the accumulators are scaled back once, in front of the flush
    const int e_dy = operand_exponent<TERMS>(p.dy_amax), e_x = operand_exponent<TERMS>(p.x_amax, p.x_amax2);
    const float dS = split_scale(e_dy), xS = split_scale(e_x);

    if (wave >= 4) {
        // =========================================== producers ===========================================
        const int pt = t - 256;
        const int lr = pt >> 2, lq = (pt & 3) * 8;           // channel row and first pixel of this thread's 8-pixel group
        struct xrow { px4<IO> a, b; px1<IO> l, r; bool ok, okl, okr; };      // x[row][x0 + lq - 1 .. x0 + lq + 8]
        struct drow { px4<IO> a, b; };

        size_t xb = 0, dyb = 0;      // element offsets into p.x / p.dy
        int x0 = 0;
        float xsc = 1.f;      // this thread's channel of p.xscale for the current unit's sample
        const int pxs = PACK ? (lq & (p.w - 1)) : lq;        // first pixel of the group inside its image row (minus x0)
        bool smp_ok = true;                                   // PACK: this group's sample exists (the last group of a batch may be short)
        auto set_unit = [&](int u) {
            const int rb = u % rblocks, sg = (u / rblocks) % segs;
            int n = u / (rblocks * segs);
            x0 = sg * SEG;
            if (PACK) { n = n * spr + lq / p.w; smp_ok = n < p.n; n = min(n, p.n - 1); }
            // inline asm like the row loads: a load the COMPILER tracks makes it put `s_waitcnt vmcnt(0)` in front of the first use of xsc in every
            // step, which drains the whole prefetch queue (measured: producers alone 1.0 ms with it, see the lab log); waited for with the prologue rows
            if (p.xscale) asm volatile("global_load_dword %0, %1, off" : "=v"(xsc) : "v"(p.xscale + (size_t)n * p.i + i0 + lr) : "memory");
            xb = ((size_t)n * p.i + i0 + lr) * plane + x0 + pxs;
            dyb = ((size_t)n * p.o + o0 + lr) * plane + x0 + pxs;
            return rb * R;
        };
        // branch-free: out-of-image rows / columns load from a clamped address and are zeroed when they are written to LDS
        auto load_x = [&](int row, xrow& r) {
            if (ABL == 7 || ABL == 11) return;
            if (ABL == 13 || ABL == 14) { r.ok = true; r.okl = r.okr = true; px4_opaque<IO>(r.a); px4_opaque<IO>(r.b); px1_opaque<IO>(r.l); px1_opaque<IO>(r.r); return; }   // opaque values instead of loads
            r.ok = row >= 0 && row < p.h;
            r.okl = r.ok && x0 + pxs - 1 >= 0;
            r.okr = r.ok && x0 + pxs + 8 < p.w;
            const size_t q = xb + (size_t)min(max(row, 0), p.h - 1) * p.w;
            px4_load<IO>(r.a, at<IO>(p.x, q));
            px4_load<IO>(r.b, at<IO>(p.x, q + 4));
            px1_load<IO>(r.l, at<IO>(p.x, q - (r.okl ? 1 : 0)));
            px1_load<IO>(r.r, at<IO>(p.x, q + (r.okr ? 8 : 7)));
        };
        auto load_dy = [&](int row, drow& r) {   // row is always inside the unit
            if (ABL == 7 || ABL == 11) return;
            if (ABL == 13 || ABL == 14) { px4_opaque<IO>(r.a); px4_opaque<IO>(r.b); return; }
            const size_t q = dyb + (size_t)row * p.w;
            px4_load<IO>(r.a, at<IO>(p.dy, q));
            px4_load<IO>(r.b, at<IO>(p.dy, q + 4));
        };
        auto touch_x = [&](xrow& r) { px4_pin<IO>(r.a); px4_pin<IO>(r.b); px1_pin<IO>(r.l); px1_pin<IO>(r.r); };
        auto touch_d = [&](drow& r) { px4_pin<IO>(r.a); px4_pin<IO>(r.b); };
        // one pair of neighbouring pixels -> packed 16-bit hi and lo (TERMS = 4: the values already carry the block scale, folded into xmul)
        auto pair = [&](float a, float b, unsigned& hi, unsigned& lo) { split2<F>(a, b, 1.f, hi, lo); };
        float xmul = 1.f;     // xsc (x block scale) for the current unit: set where xsc is known to have landed
        auto store_x = [&](int row, const xrow& r) {
            if (ABL == 7) return;
            if (ABL == 10 || ABL == 12) {   // no split arithmetic: raw register bits go to LDS
                const int slot = (row + 1) & 3;
                unsigned short* dst = xs + (size_t)slot * WS_XSLOT + lr * RS + XO + lq;
                if constexpr (IO == 0) {
                    *(u32x4*)dst = __builtin_bit_cast(u32x4, r.a.v);
                    if (TERMS > 1) *(u32x4*)(dst + 4 * WS_XSLOT) = __builtin_bit_cast(u32x4, r.b.v);
                }
                return;
            }
            float v[10];
            v[0] = r.okl ? px1_get<IO>(r.l) * xmul : 0.f;
            v[9] = r.okr ? px1_get<IO>(r.r) * xmul : 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[1 + k] = r.ok ? px4_get<IO>(r.a, k) * xmul : 0.f; v[5 + k] = r.ok ? px4_get<IO>(r.b, k) * xmul : 0.f; }
            // even-start pairs (v1v2, v3v4, v5v6, v7v8) = view 1; odd-start pairs (v0v1, ..., v8v9): view 0 = first four, view 2 = last four
            unsigned eh[4], el[4], oh[5], ol[5];
#pragma unroll
            for (int k = 0; k < 4; k++) pair(v[1 + 2 * k], v[2 + 2 * k], eh[k], el[k]);
#pragma unroll
            for (int k = 0; k < 5; k++) pair(v[2 * k], v[2 * k + 1], oh[k], ol[k]);
            const int slot = (row + 1) & 3;
            unsigned short* dst = xs + (size_t)slot * WS_XSLOT + lr * RS + XO + lq;
            if (VIEWS == 3) {
                *(u32x4*)(dst + 0 * WS_VIEW) = u32x4{oh[0], oh[1], oh[2], oh[3]};
                *(u32x4*)(dst + 1 * WS_VIEW) = u32x4{eh[0], eh[1], eh[2], eh[3]};
                *(u32x4*)(dst + 2 * WS_VIEW) = u32x4{oh[1], oh[2], oh[3], oh[4]};
                if (TERMS > 1) {
                    dst += 4 * WS_XSLOT;
                    *(u32x4*)(dst + 0 * WS_VIEW) = u32x4{ol[0], ol[1], ol[2], ol[3]};
                    *(u32x4*)(dst + 1 * WS_VIEW) = u32x4{el[0], el[1], el[2], el[3]};
                    *(u32x4*)(dst + 2 * WS_VIEW) = u32x4{ol[1], ol[2], ol[3], ol[4]};
                }
            } else {
                // one copy; the first / last thread of a row also writes the segment's left / right halo pixel (positions XROW0 - 1 and XROW0 + 32)
                *(u32x4*)dst = u32x4{eh[0], eh[1], eh[2], eh[3]};
                if (TERMS > 1) *(u32x4*)(dst + 4 * WS_XSLOT) = u32x4{el[0], el[1], el[2], el[3]};
                if (lq == 0) { dst[-1] = (unsigned short)(oh[0] & 0xffffu); if (TERMS > 1) dst[4 * WS_XSLOT - 1] = (unsigned short)(ol[0] & 0xffffu); }
                if (lq == 24) { dst[8] = (unsigned short)(oh[4] >> 16); if (TERMS > 1) dst[4 * WS_XSLOT + 8] = (unsigned short)(ol[4] >> 16); }
            }
        };
        auto store_dy = [&](int buf, const drow& r) {
            if (ABL == 7) return;
            if (ABL == 10 || ABL == 12) {
                if constexpr (IO == 0) {
                    *(u32x4*)(ds + (size_t)buf * WS_DBUF + lr * RS + lq) = __builtin_bit_cast(u32x4, r.a.v);
                    if (TERMS > 1) *(u32x4*)(ds + (size_t)(3 + buf) * WS_DBUF + lr * RS + lq) = __builtin_bit_cast(u32x4, r.b.v);
                }
                return;
            }
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = (!PACK || smp_ok) ? px4_get<IO>(r.a, k) : 0.f; v[4 + k] = (!PACK || smp_ok) ? px4_get<IO>(r.b, k) : 0.f; }
            u32x4 hi, lo;
            split8t<F>(v, dS, hi, lo);
            *(u32x4*)(ds + (size_t)buf * WS_DBUF + lr * RS + lq) = hi;
            if (TERMS > 1) *(u32x4*)(ds + (size_t)(3 + buf) * WS_DBUF + lr * RS + lq) = lo;
        };

        // Item j of a unit = {x row y0+j+2, dy row y0+j+2}: what step j writes to LDS (x for step j+1's ky = 2, dy for step j+2: dy runs two rows
        // ahead).  Its loads are issued DEPTH steps earlier and land in one of DEPTH + 1 register sets: with one step of lookahead the producers
        // alone needed 1.03-1.24 ms per layer for data the consumers use up in 0.69-0.76 ms (tools/wrw_lab.hip WRW_ABL=6 / 7; without the global
        // loads the whole kernel runs in 0.77 ms): 17 KB in flight per CU against 2-3 us of loaded-memory latency.  Every item issues the same six
        // loads (the dy row of the last item repeats the unit's last row) so that the wait counts are constants.
        constexpr int DEPTH = 3;
        struct iset { xrow x; drow d; };
        iset s0, s1, s2, s3;
        // prologue rows (x y0-1, y0, y0+1; dy y0, y0+1) of the NEXT unit are loaded during a unit's last step, when every set is idle
        xrow &px0 = s0.x, &px1 = s1.x, &px2 = s2.x;
        drow &pd0 = s0.d, &pd1 = s1.d;
        int db = 0;                        // dy buffer of the unit's first row (advances by one per row, mod 3)
        auto issue_prologue = [&](int u) {
            const int y0 = set_unit(u);
            load_x(y0 - 1, px0); load_x(y0, px1); load_x(y0 + 1, px2);
            load_dy(y0, pd0);
            if (R > 1) load_dy(y0 + 1, pd1);
            return y0;
        };
        auto load_item = [&](int y0, int j, iset& s) { load_x(y0 + j + 2, s.x); load_dy(y0 + min(j + 2, R - 1), s.d); };
        // step k of a unit (row y = y0 + k): start the loads of item k + DEPTH into `ld`, then write item k from `st`
        auto step = [&](int y0, int k, iset& ld, iset& st, int next_u) {
            if (k + DEPTH <= R - 2) load_item(y0, k + DEPTH, ld);
            if (k + 1 < R) {
                const int after = min(DEPTH, R - 2 - k);     // items issued after item k
                if (after >= 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                else if (after == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (after == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                touch_x(st.x); touch_d(st.d);
                store_x(y0 + k + 2, st.x);
                if (k + 2 < R) store_dy((db + k + 2) % 3, st.d);
            }
            if (k == R - 1 && next_u < p.units) issue_prologue(next_u);   // lands during the consumers' last row of this unit
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

        int u = split;
        if (u < p.units) issue_prologue(u);
        for (; u < p.units; u += p.splits) {
            const int y0 = set_unit(u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the prologue rows (issued during the previous unit's last step)
            touch_x(px0); touch_x(px1); touch_x(px2); touch_d(pd0); touch_d(pd1);
            asm volatile("" : "+v"(xsc));
            xmul = TERMS == 4 ? xsc * xS : xsc;
            __builtin_amdgcn_s_barrier();                              // A: the consumers are done with the previous unit
            store_x(y0 - 1, px0); store_x(y0, px1); store_x(y0 + 1, px2);
            store_dy(db % 3, pd0);
            if (R > 1) store_dy((db + 1) % 3, pd1);
            // items 0 .. DEPTH-1
            if (0 <= R - 2) load_item(y0, 0, s0);
            if (1 <= R - 2) load_item(y0, 1, s1);
            if (2 <= R - 2) load_item(y0, 2, s2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // B: rows y0-1 .. y0+1 and dy y0, y0+1 are in LDS
            const int next_u = u + p.splits;
            for (int k = 0; k < R; k += 4) {
                step(y0, k, s3, s0, next_u);
                if (k + 1 < R) step(y0, k + 1, s0, s1, next_u);
                if (k + 2 < R) step(y0, k + 2, s1, s2, next_u);
                if (k + 3 < R) step(y0, k + 3, s2, s3, next_u);
            }
            db = (db + R) % 3;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =========================================== consumers ===========================================
    const int wo = (wave >> 1) * 32, wi = (wave & 1) * 32;
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[k][e] = 0.f;
    __builtin_amdgcn_s_setprio(1);

    int db = 0;
    for (int u = split; u < p.units; u += p.splits) {
        const int y0 = ((u % rblocks)) * R;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // A
        __builtin_amdgcn_s_barrier();   // B
        asm volatile("" ::: "memory");

        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int a_lane = (wo + (ln & 31)) * RS + 8 * (ln >> 5);     // + 16 * c, bf16 units inside a dy buffer
        const int b_lane = (wi + (ln & 31)) * RS + 8 * (ln >> 5);     // + 16 * c, inside a view

        unsigned keep_l[2] = {~0u, ~0u}, keep_r[2] = {~0u, ~0u};   // PACK: per k half c, lane group g = 2c + (lane >> 5) covers pixels 8g .. 8g + 7
        if (PACK) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int px = 8 * (2 * c + (ln >> 5));
                keep_l[c] = (px & (p.w - 1)) == 0 ? 0u : ~0u;
                keep_r[c] = ((px + 8) & (p.w - 1)) == 0 ? 0u : ~0u;
            }
        }
        u32x4 a[2][2];       // [buffer][hl]
        u32x4 b[2][2][3];    // [buffer][hl][kx]   (VIEWS = 1: [..][0] is unused, [1] the aligned word, [2].x / [2].y the dwords before / after it)
        auto fetch_a = [&](int buf, int dbuf, int c) {
            a[buf][0] = *(const u32x4*)(ds + (size_t)dbuf * WS_DBUF + a_lane + 16 * c);
            if (TERMS > 1) a[buf][1] = *(const u32x4*)(ds + (size_t)(3 + dbuf) * WS_DBUF + a_lane + 16 * c);
        };
        auto fetch_b = [&](int buf, int slot, int c) {
#pragma unroll
            for (int hl = 0; hl < (TERMS > 1 ? 2 : 1); hl++) {
                const unsigned short* q = xs + (size_t)(4 * hl + slot) * WS_XSLOT + XO + b_lane + 16 * c;
                if (VIEWS == 3) {
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) b[buf][hl][kx] = *(const u32x4*)(q + kx * WS_VIEW);
                } else {
                    b[buf][hl][1] = *(const u32x4*)q;
                    b[buf][hl][2][0] = *(const unsigned*)(q - 2);
                    b[buf][hl][2][1] = *(const unsigned*)(q + 8);
                }
            }
        };
        // operand of tap kx from what fetch_b left in the registers
        auto view = [&](int buf, int hl, int kx, int c) {
            if (VIEWS == 3 || kx == 1) return b[buf][hl][kx];
            const u32x4 d = b[buf][hl][1];
            if (PACK) {   // pixel 8g - 1 / 8g + 8 of the neighbouring sample: not this sample's neighbour
                if (kx == 0) return u32x4{__builtin_amdgcn_alignbyte(d[0], b[buf][hl][2][0] & keep_l[c], 2), __builtin_amdgcn_alignbyte(d[1], d[0], 2),
                                          __builtin_amdgcn_alignbyte(d[2], d[1], 2), __builtin_amdgcn_alignbyte(d[3], d[2], 2)};
                return u32x4{__builtin_amdgcn_alignbyte(d[1], d[0], 2), __builtin_amdgcn_alignbyte(d[2], d[1], 2), __builtin_amdgcn_alignbyte(d[3], d[2], 2),
                             __builtin_amdgcn_alignbyte(b[buf][hl][2][1] & keep_r[c], d[3], 2)};
            }
            const unsigned a01 = __builtin_amdgcn_alignbyte(d[1], d[0], 2), a12 = __builtin_amdgcn_alignbyte(d[2], d[1], 2), a23 = __builtin_amdgcn_alignbyte(d[3], d[2], 2);
            if (kx == 0) return u32x4{__builtin_amdgcn_alignbyte(d[0], b[buf][hl][2][0], 2), a01, a12, a23};
            return u32x4{a01, a12, a23, __builtin_amdgcn_alignbyte(b[buf][hl][2][1], d[3], 2)};
        };
        constexpr int RB = (TERMS > 1 ? 2 : 1) * 3;   // LDS reads of one fetch_b
        constexpr int RA = TERMS > 1 ? 2 : 1;         // ... of one fetch_a
        // first sub-step of the unit's first row (nothing could be fetched ahead across barrier B)
        fetch_a(0, db % 3, 0);
        fetch_b(0, y0 & 3, 0);

        for (int k = 0; k < R; k++) {
            const int y = y0 + k;
            const int dbuf = (db + k) % 3, dnext = (db + k + 1) % 3;
            // six sub-steps j = (c, ky): operands of sub-step j+1 are fetched before the nine MFMAs of sub-step j; the last one fetches the
            // first operands of row y+1 (dy row y+1 and x row y have been in LDS since before the previous barrier)
#pragma unroll
            for (int j = 0; j < ((ABL == 6 || ABL == 12 || ABL == 13) ? 0 : 6); j++) {
                const int c = j / 3, ky = j % 3;
                const int cur = j & 1, nxt = cur ^ 1;
                int reads = RB;
                if (j < 5) {
                    const int c1 = (j + 1) / 3, ky1 = (j + 1) % 3;
                    fetch_b(nxt, (y + ky1) & 3, c1);
                    if (ky1 == 0) { fetch_a(nxt, dbuf, c1); reads += RA; }
                } else {   // unconditional (after the unit's last row it reads valid, unused LDS words): no branch inside the pinned schedule
                    fetch_b(nxt, (y + 1) & 3, 0);
                    fetch_a(nxt, dnext, 0);
                    reads += RA;
                }
                const u32x4 a_hi = a[c][0], a_lo = a[c][1];   // the dy operand of k-half c lives in buffer c
                u32x4 bh[3], bl[3];
#pragma unroll
                for (int kx = 0; kx < 3; kx++) { bh[kx] = view(cur, 0, kx, c); if (TERMS > 1) bl[kx] = view(cur, 1, kx, c); }
                if (TERMS > 1) {
#pragma unroll
                    for (int kx = 0; kx < 3; kx++)
                        acc[ky * 3 + kx] = mma16<F>(a_lo, bh[kx], acc[ky * 3 + kx]);
#pragma unroll
                    for (int kx = 0; kx < 3; kx++)
                        acc[ky * 3 + kx] = mma16<F>(a_hi, bl[kx], acc[ky * 3 + kx]);
                }
#pragma unroll
                for (int kx = 0; kx < 3; kx++)
                    acc[ky * 3 + kx] = mma16<F>(a_hi, bh[kx], acc[ky * 3 + kx]);
                constexpr int MF = TERMS > 1 ? 9 : 3;
                // RPM operand reads behind each of the first MFMAs: the earlier the last read issues, the more MFMAs cover its LDS latency
                constexpr int RPM = TERMS > 1 ? SGV_WRW_RPM : 3;
#pragma unroll
                for (int i = 0; i < MF; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                    for (int r2 = 0; r2 < RPM; r2++)
                        if (i * RPM + r2 < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        db = (db + R) % 3;
    }

    if (TERMS == 4) {     // the two block scales come off before the sums leave the registers
        const int eu = unscale_exponent(e_dy, e_x);
#pragma unroll
        for (int k = 0; k < 9; k++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[k][e] = __builtin_ldexpf(acc[k][e], eu);
    }
    // Flush (behind the last row's barrier nobody reads or writes the operand tiles any more).
    if (ABL != 8 && !p.scatter_flush) { flush_tile(acc, (float*)lds_ws + wave * FLUSH_STAGE_FLOATS, p.dw, p.i, o0 + wo, i0 + wi); return; }
    // C layout of the 32x32 MFMA: col (i) = lane & 31, row (o) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const int r32 = lane & 31, g = lane >> 5;
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int o = o0 + wo + (e & 3) + 8 * (e >> 2) + 4 * g;
            const int i = i0 + wi + r32;
            if (ABL == 8) { if (acc[k][e] == 12345.678f) atomicAdd(p.dw + ((size_t)o * p.i + i) * 9 + k, acc[k][e]); }   // ablation: (practically) no flush
            else atomicAdd(p.dw + ((size_t)o * p.i + i) * 9 + k, acc[k][e]);
        }
}

}  // namespace sgv_wrw
