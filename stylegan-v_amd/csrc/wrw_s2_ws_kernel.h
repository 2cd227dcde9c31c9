// Weight gradient of the 3x3 / stride-2 layer pair (strided: big -> small, transposed: small -> big) -- producer / consumer form of
// wrw3x3_s2_kernel (wrw_kernel.h): same arithmetic (bf16 hi/lo split products on v_mfma_f32_32x32x16_bf16, fp32 accumulate), same LDS row
// layout (even / odd column planes of the big rows), same work decomposition (64 cs x 64 cb x 9 taps per workgroup, persistent over (sample,
// 32-pixel column segment of the small grid, row block) units, LDS-staged atomic flush at the end).
//
//     dw[cs, cb, ky, kx] = sum_{n,Y,X} small[n,cs,Y,X] * big[n,cb,2Y+ky,2X+kx]
//
// Reference: `Conv2dGradWeight` of conv2d_gradfix.py:140-170 for the down-sampling convolutions of DiscriminatorBlock (networks.py:470-475 via
// conv2d_resample.py:119-122) and the up-sampling ones of SynthesisLayer (networks.py:141 via conv2d_resample.py:125-137).
//
// Why: in the 4-wave kernel every wave loads two big rows and a small row, splits them, writes ~40 LDS words and only then feeds the matrix
// pipe -- 27 % MFMA-busy in the train step (profiles/r02_pmc_bench_step_MFMA_table.txt).  Same cure as wrw_ws_kernel.h / conv3x3_ws_kernel.h:
//   waves 0-3 (consumers, 2 x 2 over the 64 x 64 tile): LDS operand reads + MFMAs only (144 accumulators); operands are fetched one (k half, ky)
//                   sub-step ahead, the first sub-step of row Y+1 before the barrier that ends row Y;
//   waves 4-7 (producers): global loads (inline asm, counted vmcnt, two register sets: the rows of step Y+1 are in flight during step Y), hi/lo
//                   split, even / odd de-interleave of the big rows.
// LDS: big ring 5 rows x {hi,lo} x 64 ch x 176 B = 110 KiB (rows 2Y .. 2Y+2 in use, 2Y+3, 2Y+4 being written), small 3 rows x {hi,lo} x 64 ch x 80 B
// = 30 KiB (the small operand runs two rows ahead so that the next step's first operands are readable before the barrier): 140 KiB, one workgroup per CU.
// A step moves 41 KB per CU (two big rows + one small row) for 54 MFMAs: 24 B per MFMA-pipe cycle, above what a CU's memory path sustains (~20 B) --
// the kernel is bound by its producers, not by the matrix pipe.
#pragma once

#include "wrw_kernel.h"
#include "sgv_io16.h"

namespace sgv_wrw {

constexpr int S2W_SMALL = TO * RS;                                                // bf16 per (buffer, hl) of the small operand
constexpr int WRW_S2_WS_LDS_BYTES = (2 * 5 * BIG_SLOT + 2 * 3 * S2W_SMALL) * 2;

// PACK: as in wrw3x3_s2_kernel (small grid 16 / 8 pixels wide, 2 / 4 samples per row step, per-sample last big column in the row's pad words).
// ABL (tools/wrw_lab.hip only; wrong results by construction): 6 consumers only keep the barrier protocol, 7 producers only keep it, 9 full kernel with the
// big-row loads forced onto 16-byte boundaries (for odd b the rows then start 12 bytes early: still 4-byte granular, see the lab log).
// IO: element format of the two activation tensors (sgv_io16.h; 16-bit tensors with TERMS = 1; dw stays fp32): the same load instructions at half the width.
template <int TERMS, bool PACK = false, int ABL = 0, int IO = 0>
__global__ __launch_bounds__(512, 2) void wrw3x3_s2_ws_kernel(wrw_s2_params p) {
    constexpr int F = sgv_conv::operand_format<TERMS, IO>();      // operand format of the products (sgv_split.h): TERMS, or 2 = fp16 operands for fp16 tensors
    using namespace sgv_io;
    static_assert(IO == 0 || TERMS == 1, "16-bit tensors are single bf16 operands");
    constexpr int ES = fmt<IO>::ES;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_s2w[];
    unsigned short* bs = lds_s2w;                          // ((hl * 5 + slot) * 64 + cb) * BIG_CH + {px: even plane | RS + px: odd plane | 2 RS ..: pad}
    unsigned short* as = lds_s2w + 2 * 5 * BIG_SLOT;       // ((hl * 3 + buf) * 64 + cs) * RS + px

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int vid = (nwg & 7) == 0 ? (lin & 7) * (nwg >> 3) + (lin >> 3) : lin;
    const int tile = vid % (int)gridDim.x, split = vid / (int)gridDim.x;
    const int s0 = (tile / p.tiles_b) * TO, b0 = (tile % p.tiles_b) * TI;
    const int segs = PACK ? 1 : p.w / SEG, rblocks = p.h / p.rows;
    const int wsh = PACK ? 31 - __builtin_clz(p.w) : 5;   // log2(W): W is 16 or 8 when PACK
    const int spr = PACK ? SEG >> wsh : 1;                // samples per row step
    const int hb = 2 * p.h + 1, wb = 2 * p.w + 1;
    const size_t plane_s = (size_t)p.h * p.w, plane_b = (size_t)hb * wb;
    const int R = p.rows;
    // TERMS = 4: block exponents of the two operands (sgv_split.h); the accumulators are scaled back once, in front of the flush
    const int e_s = operand_exponent<TERMS>(p.small_amax), e_b = operand_exponent<TERMS>(p.big_amax);
    const float sS = split_scale(e_s), bS = split_scale(e_b);

    if (wave >= 4) {
        // =========================================== producers ===========================================
        const int pt = t - 256;
        const int lr = pt >> 2, lq = (pt & 3) * 8;     // small row: channel, first pixel of this thread's 8-pixel group
        struct bset { px4<IO> v[4]; px1<IO> e; };         // one big row: four 4-column quads (quad (pt + 256 j) & 15 of channel (pt + 256 j) >> 4) + this thread's edge column
        struct sset { px4<IO> a, b; };                // one small row: 8 pixels
        struct rset { bset b0, b1; sset s; };

        const char* bq[4] = {nullptr, nullptr, nullptr, nullptr};    // first element of this thread's four quads in local big row 0
        const char* be = nullptr;                                      // ... of its edge column (channel pt & 63; PACK: of sample pt >> 6)
        const char* sq = nullptr;                                      // ... of its 8 small pixels in local small row 0
        bool live = true;                                              // PACK: this thread's small sample exists (the last group of a batch may be short)
        auto set_unit = [&](int u) {
            const int rb = u % rblocks, sg = (u / rblocks) % segs, n = (u / (rblocks * segs)) * spr;
            const int y0 = rb * R, x0 = sg * SEG;
            const char* bb = at<IO>(p.big, ((size_t)n * p.cb + b0) * plane_b + (size_t)(2 * y0) * wb + 2 * x0);
            const char* sb = at<IO>(p.small, ((size_t)n * p.cs + s0) * plane_s + (size_t)y0 * p.w + x0);
            const int last = p.n - 1 - n;     // samples of the group beyond the batch are clamped to its last one (their small operand is zeroed)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int it = pt + 256 * j, quad = it & 15, ch = it >> 4;
                if (PACK) bq[j] = bb + (((size_t)min(quad >> (wsh - 1), last) * p.cb + ch) * plane_b + 4 * (quad & ((p.w >> 1) - 1))) * ES;
                else bq[j] = bb + ((size_t)ch * plane_b + 4 * quad) * ES;
            }
            if (ABL == 9) {   // lab: what would 16-byte aligned big rows buy (wrong data)
#pragma unroll
                for (int j = 0; j < 4; j++) bq[j] = (const char*)((uintptr_t)bq[j] & ~(uintptr_t)15);
            }
            if (PACK) be = bb + (((size_t)min(min(pt >> 6, spr - 1), last) * p.cb + (pt & 63)) * plane_b + 2 * p.w) * ES;
            else be = bb + ((size_t)(pt & 63) * plane_b + 64) * ES;
            if (PACK) {
                const int smp = lq >> wsh;
                live = smp <= last;
                sq = sb + (((size_t)min(smp, last) * p.cs + lr) * plane_s + (lq & (p.w - 1))) * ES;
            } else sq = sb + ((size_t)lr * plane_s + lq) * ES;
        };
        // The loads are inline asm (counted s_waitcnt below; see conv3x3_ws_kernel.h): their destinations are unprotected until `touch`.
        auto load_big = [&](int b, bset& r) {   // local big row b = 0 .. 2 R
            if (ABL == 6) return;
            const size_t o = (size_t)b * wb * ES;
#pragma unroll
            for (int j = 0; j < 4; j++) px4_load<IO>(r.v[j], bq[j] + o);
            px1_load<IO>(r.e, be + o);
        };
        auto load_small = [&](int row, sset& r) {
            if (ABL == 6) return;
            const char* q = sq + (size_t)row * p.w * ES;
            px4_load<IO>(r.a, q);
            px4_load<IO>(r.b, q + 4 * ES);
        };
        auto touch_b = [&](bset& r) {
#pragma unroll
            for (int j = 0; j < 4; j++) px4_pin<IO>(r.v[j]);
            px1_pin<IO>(r.e);
        };
        auto touch_s = [&](sset& r) { px4_pin<IO>(r.a); px4_pin<IO>(r.b); };
        auto store_big = [&](int b, const bset& r) {
            if (ABL == 6) return;
            const int slot = b - 5 * ((b * 205) >> 10);   // b % 5
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int it = pt + 256 * j, quad = it & 15, ch = it >> 4;
                const float v0 = px4_get<IO>(r.v[j], 0), v1 = px4_get<IO>(r.v[j], 1), v2 = px4_get<IO>(r.v[j], 2), v3 = px4_get<IO>(r.v[j], 3);
                unsigned he, ho, le, lo;
                split2<F>(v0, v2, bS, he, le);     // even columns
                split2<F>(v1, v3, bS, ho, lo);     // odd columns
                const int pos = slot * BIG_SLOT + ch * BIG_CH + 2 * quad;
                *(unsigned*)&bs[pos] = he;
                *(unsigned*)&bs[pos + RS] = ho;
                if (TERMS > 1) {
                    *(unsigned*)&bs[5 * BIG_SLOT + pos] = le;
                    *(unsigned*)&bs[5 * BIG_SLOT + pos + RS] = lo;
                }
            }
            if (pt < TI * spr) {
                const float ev = px1_get<IO>(r.e);
                unsigned h, l;
                split2<F>(ev, 0.f, bS, h, l);
                const int pos = slot * BIG_SLOT + (pt & 63) * BIG_CH + (PACK ? 2 * RS + 2 * (pt >> 6) : 32);
                bs[pos] = (unsigned short)h;
                if (TERMS > 1) bs[5 * BIG_SLOT + pos] = (unsigned short)l;
            }
        };
        auto store_small = [&](int buf, const sset& r) {
            if (ABL == 6) return;
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = (!PACK || live) ? px4_get<IO>(r.a, k) : 0.f; v[4 + k] = (!PACK || live) ? px4_get<IO>(r.b, k) : 0.f; }
            u32x4 hi, lo;
            split8t<F>(v, sS, hi, lo);
            *(u32x4*)&as[buf * S2W_SMALL + lr * RS + lq] = hi;
            if (TERMS > 1) *(u32x4*)&as[(3 + buf) * S2W_SMALL + lr * RS + lq] = lo;
        };

        rset sa, sb_;            // step sets
        // prologue rows (big 0, 1, 2; small 0, 1) of the NEXT unit are loaded during a unit's last step, when both step sets are idle (step R-1 neither
        // loads nor stores): they live in the same registers
        bset &pb0 = sa.b0, &pb1 = sa.b1, &pb2 = sb_.b0;
        sset &ps0 = sa.s, &ps1 = sb_.s;
        int db = 0;              // small buffer of the unit's first row (advances by one per row, mod 3)
        auto issue_prologue = [&](int u) {
            set_unit(u);
            load_big(0, pb0); load_big(1, pb1); load_big(2, pb2);
            load_small(0, ps0);
            if (R > 1) load_small(1, ps1);
        };
        // step k of a unit (small row k, big rows 2k .. 2k+2): start the loads of big rows 2k+5, 2k+6 / small row k+3 into `ld`, then write big rows
        // 2k+3, 2k+4 (step k+1's) and small row k+2 (step k+2's: the small operand runs two rows ahead) from `st`, whose loads were issued a step ago.
        auto step = [&](int k, rset& ld, rset& st, int next_u) {
            const bool issue = k + 2 < R;
            if (issue) { load_big(2 * k + 5, ld.b0); load_big(2 * k + 6, ld.b1); if (k + 3 < R) load_small(k + 3, ld.s); }
            if (k + 1 < R) {
                if (issue) { if (k + 3 < R) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                touch_b(st.b0); touch_b(st.b1);
                store_big(2 * k + 3, st.b0);
                store_big(2 * k + 4, st.b1);
                if (k + 2 < R) { touch_s(st.s); store_small((db + k + 2) % 3, st.s); }
            }
            if (k == R - 1 && next_u < p.units) issue_prologue(next_u);   // lands during the consumers' last row of this unit
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

        int u = split;
        if (u < p.units) issue_prologue(u);
        for (; u < p.units; u += p.splits) {
            set_unit(u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the prologue rows (issued during the previous unit's last step)
            touch_b(pb0); touch_b(pb1); touch_b(pb2); touch_s(ps0); touch_s(ps1);
            __builtin_amdgcn_s_barrier();                              // A: the consumers are done with the previous unit
            store_big(0, pb0); store_big(1, pb1); store_big(2, pb2);
            store_small(db % 3, ps0);
            if (R > 1) store_small((db + 1) % 3, ps1);
            // the set of step 0: big rows 3, 4 and small row 2
            if (R > 1) { load_big(3, sa.b0); load_big(4, sa.b1); if (R > 2) load_small(2, sa.s); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // B: big rows 0 .. 2 and small rows 0, 1 are in LDS
            const int next_u = u + p.splits;
            for (int k = 0; k < R; k += 2) {
                step(k, sb_, sa, next_u);
                if (k + 1 < R) step(k + 1, sa, sb_, next_u);
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // A
        __builtin_amdgcn_s_barrier();   // B
        asm volatile("" ::: "memory");

        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int a_lane = (wo + (ln & 31)) * RS + 8 * (ln >> 5);        // + 16 c, bf16 units inside a small buffer
        const int b_lane = (wi + (ln & 31)) * BIG_CH + 8 * (ln >> 5);    // + 16 c, inside a ring slot: even plane; + RS: odd plane
        // the dword behind this lane's eight even columns (kx = 2 reads the even plane one column to the right); PACK: where the group ends a sample,
        // that sample's last big column (kept in the pad words)
        int e_lane[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int px = 16 * c + 8 * (ln >> 5);
            int off = px + 8;
            if (PACK && ((px + 8) & (p.w - 1)) == 0) off = 2 * RS + 2 * (px >> wsh);
            e_lane[c] = (wi + (ln & 31)) * BIG_CH + off;
        }

        u32x4 a[2][2];                                   // [buffer][hl]
        struct bop { u32x4 e, o; unsigned ea; };
        bop b[2][2];                                     // [buffer][hl]
        auto fetch_a = [&](int buf, int sbuf, int c) {
            a[buf][0] = *(const u32x4*)(as + (size_t)sbuf * S2W_SMALL + a_lane + 16 * c);
            if (TERMS > 1) a[buf][1] = *(const u32x4*)(as + (size_t)(3 + sbuf) * S2W_SMALL + a_lane + 16 * c);
        };
        auto fetch_b = [&](int buf, int slot, int c) {
#pragma unroll
            for (int hl = 0; hl < (TERMS > 1 ? 2 : 1); hl++) {
                const unsigned short* q = bs + (size_t)(5 * hl + slot) * BIG_SLOT;
                b[buf][hl].e = *(const u32x4*)(q + b_lane + 16 * c);
                b[buf][hl].o = *(const u32x4*)(q + b_lane + 16 * c + RS);
                b[buf][hl].ea = *(const unsigned*)(q + e_lane[c]);
            }
        };
        auto view = [&](int buf, int hl, int kx) {
            if (kx == 0) return b[buf][hl].e;
            if (kx == 1) return b[buf][hl].o;
            const u32x4 d = b[buf][hl].e;
            return u32x4{__builtin_amdgcn_alignbyte(d[1], d[0], 2), __builtin_amdgcn_alignbyte(d[2], d[1], 2), __builtin_amdgcn_alignbyte(d[3], d[2], 2),
                         __builtin_amdgcn_alignbyte(b[buf][hl].ea, d[3], 2)};
        };
        constexpr int RB = (TERMS > 1 ? 2 : 1) * 3;   // LDS reads of one fetch_b
        constexpr int RA = TERMS > 1 ? 2 : 1;         // ... of one fetch_a
        // first sub-step of the unit's first row (nothing could be fetched ahead across barrier B)
        fetch_a(0, db % 3, 0);
        fetch_b(0, 0, 0);

        int s_base = 0;    // ring slot of local big row 2k
        for (int k = 0; k < R; k++) {
            const int sbuf = (db + k) % 3, snext = (db + k + 1) % 3;
            // six sub-steps j = (c, ky): operands of sub-step j+1 are fetched before the nine MFMAs of sub-step j; the last one fetches the first
            // operands of row k+1 (small row k+1 and big row 2k+2 have been in LDS since before the previous barrier)
#pragma unroll
            for (int j = 0; j < (ABL == 7 ? 0 : 6); j++) {
                const int c = j / 3, ky = j % 3;
                const int cur = j & 1, nxt = cur ^ 1;
                int reads = RB;
                if (j < 5) {
                    const int c1 = (j + 1) / 3, ky1 = (j + 1) % 3;
                    const int sl = s_base + ky1;
                    fetch_b(nxt, sl >= 5 ? sl - 5 : sl, c1);
                    if (ky1 == 0) { fetch_a(nxt, sbuf, c1); reads += RA; }
                } else {   // unconditional (after the unit's last row it reads valid, unused LDS words): no branch inside the pinned schedule
                    const int sl = s_base + 2;
                    fetch_b(nxt, sl >= 5 ? sl - 5 : sl, 0);
                    fetch_a(nxt, snext, 0);
                    reads += RA;
                }
                const u32x4 a_hi = a[c][0], a_lo = a[c][1];   // the small operand of k-half c lives in buffer c
                u32x4 bh[3], bl[3];
#pragma unroll
                for (int kx = 0; kx < 3; kx++) { bh[kx] = view(cur, 0, kx); if (TERMS > 1) bl[kx] = view(cur, 1, kx); }
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
                constexpr int RPM = TERMS > 1 ? 2 : 3;   // operand reads behind each of the first MFMAs
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
            s_base = s_base + 2 >= 5 ? s_base - 3 : s_base + 2;
        }
        db = (db + R) % 3;
    }

    if (TERMS == 4) {     // the two block scales come off before the sums leave the registers
        const int eu = unscale_exponent(e_s, e_b);
#pragma unroll
        for (int k = 0; k < 9; k++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[k][e] = __builtin_ldexpf(acc[k][e], eu);
    }
    // Flush (behind the last row's barrier nobody reads or writes the operand tiles any more).
    flush_tile(acc, (float*)lds_s2w + wave * FLUSH_STAGE_FLOATS, p.dw, p.cb, s0 + wo, b0 + wi);
}

}  // namespace sgv_wrw
