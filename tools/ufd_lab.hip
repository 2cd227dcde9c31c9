// Kernel laboratory for the upfirdn2d FIR hot call ([N,64,257,257] -> [N,64,256,256], fp32, 4x4 taps, pad 1).
// Standalone (no torch): hipcc --offload-arch=gfx950 -O3 tools/ufd_lab.hip -o ufd_lab && ./ufd_lab [N]
// Times design variants with HIP events and checks each against a naive per-output kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include <dlfcn.h>
#include "../include/sgv_ops.h"
// the product's own translation units: its kernels can then be launched directly, without the C ABI in between (V7 below)
#include "../stylegan-v_amd/csrc/sgv_runtime.hip"
#include "../stylegan-v_amd/csrc/upfirdn2d.hip"
#pragma clang fp contract(off)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct P { const float* x; const float* f; float* y; int in_w, in_h, out_w, out_h, planes, pad; float gain; int strip_h; };

// ---------------- reference: one lane per output ----------------
__global__ void k_naive(P p) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)p.planes * p.out_h * p.out_w;
    if (idx >= total) return;
    int ox = idx % p.out_w; long r = idx / p.out_w; int oy = r % p.out_h; int pl = r / p.out_h;
    const float* xp = p.x + (size_t)pl * p.in_h * p.in_w;
    float v = 0.f;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
        int iy = oy - p.pad + j, ix = ox - p.pad + i;
        float xv = (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) ? xp[(size_t)iy * p.in_w + ix] : 0.f;
        v = __builtin_fmaf(xv, p.f[(3 - j) * 4 + (3 - i)], v);
    }
    p.y[idx] = v * p.gain;
}

// ---------------- copy ceiling: float4 in -> float4 out, same bytes ----------------
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

typedef float fv4 __attribute__((ext_vector_type(4)));
template <int NTL, int NTS, int U>
__global__ __launch_bounds__(256) void k_copy2(const fv4* __restrict__ a, fv4* __restrict__ b, long n4) {
    // each block handles a contiguous chunk of U*256 float4; U loads in flight per lane
    long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
    fv4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { long i = base + u * 256; if (i < n4) v[u] = NTL ? __builtin_nontemporal_load(a + i) : a[i]; }
#pragma unroll
    for (int u = 0; u < U; u++) { long i = base + u * 256; if (i < n4) { if (NTS) __builtin_nontemporal_store(v[u], b + i); else b[i] = v[u]; } }
}

// ---------------- V1: lane-per-column walker, DPP neighbour exchange ----------------
// A wave owns 64 consecutive output columns of a strip of rows of one plane.  Per input row each lane loads ONE
// float (its tap-0 column) -- 256 B contiguous per wave, any alignment -- plus lanes 0..2 load the 3 halo columns to
// the right; taps 1..3 come from the neighbour lanes through wave_shl:1 DPP moves.  4-row sliding window in VGPRs.
__device__ __forceinline__ float wave_shl1(float v, float fill63) {
    // lane i <- lane i+1 ; lane 63 <- fill63
    int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill63), __builtin_bit_cast(int, v), 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}

template <int PF>
__global__ __launch_bounds__(256) void k_cols(P p, int colgroups, int strips) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = wave % strips;
    const int cg = (wave / strips) % colgroups;
    const int pl = wave / (strips * colgroups);
    if (pl >= p.planes) return;
    float ff[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) ff[a][b] = p.f[(3 - a) * 4 + (3 - b)];
    const float* xp = p.x + (size_t)pl * p.in_h * p.in_w;
    float* yp = p.y + (size_t)pl * p.out_h * p.out_w;
    const int ox = cg * 64 + lane;
    const int ix0 = ox - p.pad;                  // tap-0 input column of this lane
    const int ixh = cg * 64 + 64 - p.pad + lane; // halo column (lanes 0..2)
    const bool main_ok = ix0 >= 0 && ix0 < p.in_w;
    const bool halo_ok = lane < 3 && ixh < p.in_w && ixh >= 0;
    const int oy_a = strip * p.strip_h, oy_b = min(oy_a + p.strip_h, p.out_h);
    const int iy0 = oy_a - p.pad;

    float win[4][4];
    auto load_row = [&](int iy, float* dst) {
        float m = 0.f, h = 0.f;
        if (iy >= 0 && iy < p.in_h) {
            const float* row = xp + (size_t)iy * p.in_w;
            if (main_ok) m = row[ix0];
            if (halo_ok) h = row[ixh];
        }
        // taps 1..3 from the lanes to the right; lane 63 / 62 / 61 pull from the halo values held by lanes 0..2
        float h0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h), 0));
        float h1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h), 1));
        float h2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h), 2));
        dst[0] = m;
        dst[1] = wave_shl1(dst[0], h0);
        dst[2] = wave_shl1(dst[1], h1);
        dst[3] = wave_shl1(dst[2], h2);
    };
#pragma unroll
    for (int r = 0; r < 3; r++) load_row(iy0 + r, win[1 + r]);
    int iy = iy0 + 3;
    const bool st_ok = ox < p.out_w;
#pragma unroll 4
    for (int oy = oy_a; oy < oy_b; oy++) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) win[r][i] = win[r + 1][i];
        load_row(iy++, win[3]);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc = __builtin_fmaf(win[j][i], ff[j][i], acc);
        if (st_ok) yp[(size_t)oy * p.out_w + ox] = acc * p.gain;
    }
}

// ---------------- V2: lane owns 2 columns (8 B loads/stores), DPP exchange of pairs ----------------
// Same idea at twice the bytes per memory instruction: lane loads x[c], x[c+1] (taps 0,1 of its first output) and
// needs x[c+2..c+4]: the neighbour's pair plus the neighbour's neighbour's first element.
__global__ __launch_bounds__(256) void k_cols2(P p, int colgroups, int strips) {
    typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = wave % strips;
    const int cg = (wave / strips) % colgroups;
    const int pl = wave / (strips * colgroups);
    if (pl >= p.planes) return;
    float ff[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) ff[a][b] = p.f[(3 - a) * 4 + (3 - b)];
    const float* xp = p.x + (size_t)pl * p.in_h * p.in_w;
    float* yp = p.y + (size_t)pl * p.out_h * p.out_w;
    const int ox = cg * 128 + lane * 2;
    const int ix0 = ox - p.pad;
    const int ixh = cg * 128 + 128 - p.pad + lane * 2;  // halo pair (lanes 0..1 -> 4 columns, 3 needed)
    const bool main_vec = ix0 >= 0 && ix0 + 1 < p.in_w;
    const bool halo_vec = lane < 2 && ixh >= 0 && ixh + 1 < p.in_w;
    const int oy_a = strip * p.strip_h, oy_b = min(oy_a + p.strip_h, p.out_h);
    const int iy0 = oy_a - p.pad;
    float win[4][5];
    auto load_row = [&](int iy, float* dst) {
        float m0 = 0.f, m1 = 0.f, g0 = 0.f, g1 = 0.f;
        if (iy >= 0 && iy < p.in_h) {
            const float* row = xp + (size_t)iy * p.in_w;
            if (main_vec) { f2 v = *(const f2*)(row + ix0); m0 = v[0]; m1 = v[1]; }
            else { if (ix0 >= 0 && ix0 < p.in_w) m0 = row[ix0]; if (ix0 + 1 >= 0 && ix0 + 1 < p.in_w) m1 = row[ix0 + 1]; }
            if (halo_vec) { f2 v = *(const f2*)(row + ixh); g0 = v[0]; g1 = v[1]; }
            else if (lane < 2) { if (ixh >= 0 && ixh < p.in_w) g0 = row[ixh]; if (ixh + 1 >= 0 && ixh + 1 < p.in_w) g1 = row[ixh + 1]; }
        }
        float h0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g0), 0));
        float h1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g1), 0));
        float h2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g0), 1));
        dst[0] = m0; dst[1] = m1;
        dst[2] = wave_shl1(m0, h0);       // x[c+2] = next lane's m0
        dst[3] = wave_shl1(m1, h1);       // x[c+3] = next lane's m1
        dst[4] = wave_shl1(dst[2], h2);   // x[c+4] = next-next lane's m0
    };
#pragma unroll
    for (int r = 0; r < 3; r++) load_row(iy0 + r, win[1 + r]);
    int iy = iy0 + 3;
#pragma unroll 4
    for (int oy = oy_a; oy < oy_b; oy++) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int i = 0; i < 5; i++) win[r][i] = win[r + 1][i];
        load_row(iy++, win[3]);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) { a0 = __builtin_fmaf(win[j][i], ff[j][i], a0); a1 = __builtin_fmaf(win[j][i + 1], ff[j][i], a1); }
        float* yr = yp + (size_t)oy * p.out_w + ox;
        if (ox + 1 < p.out_w) { f2 o; o[0] = a0 * p.gain; o[1] = a1 * p.gain; *(f2*)yr = o; }
        else if (ox < p.out_w) yr[0] = a0 * p.gain;
    }
}


// ---------------- V3: lane owns 4 columns (16 B unaligned loads / aligned stores), DPP halo, explicit prefetch ----------------
// Loads are always in-bounds vector loads (base clamped into the row, edge lanes fixed up with selects: no divergent
// slow path); the 3 halo columns of the wave come from one masked dword load by lanes 0..2; rows are software-pipelined
// PF deep (loads of the next PF rows in flight while the current PF rows are filtered).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4), aligned(16)));

struct rawrow { float m[4]; float h; };

template <int PF, int NT, int ABL = 0>
__global__ __launch_bounds__(256) void k_cols4(P p, int colgroups, int strips) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = wave % strips;
    const int cg = (wave / strips) % colgroups;
    const int pl = wave / (strips * colgroups);
    if (pl >= p.planes) return;
    float ff[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) ff[a][b] = p.f[(3 - a) * 4 + (3 - b)];
    const float* xp = p.x + (size_t)pl * p.in_h * p.in_w;
    float* yp = p.y + (size_t)pl * p.out_h * p.out_w;
    const int ox = cg * 256 + lane * 4;
    const int ix0 = ox - p.pad;
    // clamp the 4-wide load window into [0, in_w-4]; `sh` = how far it moved (positive: window moved right)
    const int base = min(max(ix0, 0), p.in_w - 4);
    const int sh = base - ix0;      // in [-3, 3] for edge lanes, 0 inside
    const bool lane_dead = (ix0 >= p.in_w) || (ix0 + 3 < 0);
    const int ixh = cg * 256 + 256 - p.pad + lane;
    const bool halo_ok = lane < 3 && ixh >= 0 && ixh < p.in_w;
    const int oy_a = strip * p.strip_h, oy_b = min(oy_a + p.strip_h, p.out_h);
    const int iy0 = oy_a - p.pad;

    auto issue = [&](int iy, rawrow& r) {
        r.m[0] = r.m[1] = r.m[2] = r.m[3] = 0.f; r.h = 0.f;
        if (iy >= 0 && iy < p.in_h) {   // wave-uniform
            const float* row = xp + (size_t)iy * p.in_w;
            if (!lane_dead) { f4u v = (NT == 2) ? __builtin_nontemporal_load((const f4u*)(row + base)) : *(const f4u*)(row + base); r.m[0] = v[0]; r.m[1] = v[1]; r.m[2] = v[2]; r.m[3] = v[3]; }
            if (halo_ok) r.h = row[ixh];
        }
    };
    auto expand = [&](const rawrow& r, float* dst) {
        float m[4];
        // undo the clamp: m[i] = column ix0+i = loaded[(i - sh)] if 0 <= i - sh < 4 else 0
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = r.m[i];
#pragma unroll
            for (int d = 1; d <= 3; d++) {
                if (i - d >= 0) v = (sh == d) ? r.m[i - d] : v;
                else v = (sh == d) ? 0.f : v;
                if (i + d < 4) v = (sh == -d) ? r.m[i + d] : v;
                else v = (sh == -d) ? 0.f : v;
            }
            m[i] = v;
        }
        float h0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.h), 0));
        float h1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.h), 1));
        float h2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.h), 2));
        dst[0] = m[0]; dst[1] = m[1]; dst[2] = m[2]; dst[3] = m[3];
        dst[4] = wave_shl1(m[0], h0);
        dst[5] = wave_shl1(m[1], h1);
        dst[6] = wave_shl1(m[2], h2);
    };

    float win[4][7];
    rawrow cur[PF], nxt[PF];
    {   // prologue: 3 window rows, then the first PF raw rows
        rawrow t;
#pragma unroll
        for (int r = 0; r < 3; r++) { issue(iy0 + r, t); expand(t, win[1 + r]); }
    }
    int iy = iy0 + 3;
#pragma unroll
    for (int k = 0; k < PF; k++) issue(iy + k, cur[k]);
    iy += PF;
    const bool st_vec = ox + 3 < p.out_w;
    for (int oy = oy_a; oy < oy_b; oy += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) issue(iy + k, nxt[k]);   // next group's loads go out before this group's math
        iy += PF;
#pragma unroll
        for (int k = 0; k < PF; k++) {
            if (oy + k >= oy_b) break;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int i = 0; i < 7; i++) win[r][i] = win[r + 1][i];
            if (ABL != 2) expand(cur[k], win[3]);
            float o[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                float acc = 0.f;
                if (ABL == 2) { o[v] = cur[k].m[v] + cur[k].h; continue; }                       // ablation: loads + stores only
                if (ABL == 1) { o[v] = win[3][v] + win[3][v + 3] + win[0][v]; continue; }         // ablation: no FMAs (window and DPP kept)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc = __builtin_fmaf(win[j][v + i], ff[j][i], acc);
                o[v] = acc * p.gain;
            }
            float* yr = yp + (size_t)(oy + k) * p.out_w + ox;
            if (st_vec) {
                f4u sv; sv[0] = o[0]; sv[1] = o[1]; sv[2] = o[2]; sv[3] = o[3];
                if (NT) __builtin_nontemporal_store(sv, (f4u*)yr); else *(f4u*)yr = sv;
            } else {
#pragma unroll
                for (int v = 0; v < 4; v++) if (ox + v < p.out_w) yr[v] = o[v];
            }
        }
#pragma unroll
        for (int k = 0; k < PF; k++) cur[k] = nxt[k];
    }
}

// ---------------- V4: V3 with the row loads as inline asm and counted waits ----------------
// hipcc puts `s_waitcnt vmcnt(0)` directly behind the row load of V3 / of the product kernel (the loaded value is merged with the zero row through a
// select inside the same conditional block), so no load is in flight while the previous rows are filtered: only occupancy hides the memory
// latency.  Here every row issues exactly two loads (main 16 B + halo dword; out-of-range rows load a clamped row and are zeroed on use) and
// the wait in front of a row group's use counts the VMEM instructions issued behind its loads (the next group's loads are issued BEFORE the wait:
// two groups in flight).  Requires whole-vector stores in every lane (out_w % 4 == 0) and in_w >= 4.
template <int PF>
__global__ __launch_bounds__(256) void k_cols4_asm(P p, int colgroups, int strips, int getenv_drain) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = wave % strips;
    const int cg = (wave / strips) % colgroups;
    const int pl = wave / (strips * colgroups);
    if (pl >= p.planes) return;
    float ff[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) ff[a][b] = p.f[(3 - a) * 4 + (3 - b)];
    const float* xp = p.x + (size_t)pl * p.in_h * p.in_w;
    float* yp = p.y + (size_t)pl * p.out_h * p.out_w;
    const int ox = cg * 256 + lane * 4;
    const int ix0 = ox - p.pad;
    const int base = min(max(ix0, 0), p.in_w - 4);
    const int sh = base - ix0;
    const bool lane_dead = (ix0 >= p.in_w) || (ix0 + 3 < 0);
    const int ixh = cg * 256 + 256 - p.pad + lane;
    const bool halo_ok = lane < 3 && ixh >= 0 && ixh < p.in_w;
    const int ixh_c = min(max(ixh, 0), p.in_w - 1);
    const int oy_a = strip * p.strip_h, oy_b = min(oy_a + p.strip_h, p.out_h);
    const int iy0 = oy_a - p.pad;
    typedef float f4v __attribute__((ext_vector_type(4)));
    struct raw { f4v m; float h; };
    auto issue = [&](int iy, raw& r) {   // always two loads
        const float* row = xp + (size_t)min(max(iy, 0), p.in_h - 1) * p.in_w;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.m) : "v"(row + base) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(r.h) : "v"(row + ixh_c) : "memory");
    };
    auto expand = [&](int iy, raw& r, float* dst) {
        asm volatile("" : "+v"(r.m)); asm volatile("" : "+v"(r.h));
        const bool row_ok = iy >= 0 && iy < p.in_h;
        float m[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = r.m[i];
#pragma unroll
            for (int d = 1; d <= 3; d++) {
                if (i - d >= 0) v = (sh == d) ? r.m[i - d] : v; else v = (sh == d) ? 0.f : v;
                if (i + d < 4) v = (sh == -d) ? r.m[i + d] : v; else v = (sh == -d) ? 0.f : v;
            }
            m[i] = (row_ok && !lane_dead) ? v : 0.f;
        }
        const float hv = (row_ok && halo_ok) ? r.h : 0.f;
        float h0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), 0));
        float h1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), 1));
        float h2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), 2));
        dst[0] = m[0]; dst[1] = m[1]; dst[2] = m[2]; dst[3] = m[3];
        dst[4] = wave_shl1(m[0], h0);
        dst[5] = wave_shl1(m[1], h1);
        dst[6] = wave_shl1(m[2], h2);
    };
    float win[4][7];
    // Two register sets used alternately and never copied: the compiler believes an asm load's result is there at once, so nothing may read, move
    // or re-allocate these registers between the load and the `touch` behind the counted wait (expand() starts with it).
    raw pr[3], sa[PF], sb[PF];
#pragma unroll
    for (int r = 0; r < 3; r++) issue(iy0 + r, pr[r]);
    int iy = iy0 + 3;
#pragma unroll
    for (int k = 0; k < PF; k++) issue(iy + k, sa[k]);
    if (PF == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (PF == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (PF == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 3; r++) expand(iy0 + r, pr[r], win[1 + r]);
    auto group = [&](int oy, raw* cur, raw* nxt) {
#pragma unroll
        for (int k = 0; k < PF; k++) issue(iy + PF + k, nxt[k]);   // the group after this one
        // Newer than this group's loads: the previous group's PF stores and the 2 PF loads just issued.  Loads complete in order among themselves, but
        // a store can be acknowledged before an older load has returned (waiting for <= 3 PF outstanding gave wrong results: 2-8 M mismatches), so
        // the only safe count is the number of newer LOADS -- which also makes the wave wait for the previous group's stores.
        if (getenv_drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (PF == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (PF == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (PF == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
        for (int k = 0; k < PF; k++) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int i = 0; i < 7; i++) win[r][i] = win[r + 1][i];
            expand(iy + k, cur[k], win[3]);
            float o[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc = __builtin_fmaf(win[j][v + i], ff[j][i], acc);
                o[v] = acc * p.gain;
            }
            float* yr = yp + (size_t)(oy + k) * p.out_w + ox;
            f4v sv; sv[0] = o[0]; sv[1] = o[1]; sv[2] = o[2]; sv[3] = o[3];
            // s_nop: a store of more than 64 bits reads its data registers a few cycles AFTER issue; the compiler's hazard recogniser inserts the wait
            // states for its own stores but cannot see into inline asm (without it the next address computation landed in the data registers:
            // 15 wrong outputs per row group)
            asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 2" :: "v"(yr), "v"(sv) : "memory");
        }
        iy += PF;
    };
    for (int oy = oy_a; oy < oy_b; oy += 2 * PF) {     // strip heights are multiples of 2 PF
        group(oy, sa, sb);
        group(oy + PF, sb, sa);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---------------- V6: tile through LDS, every load of the workgroup issued up front (copy-like request pattern) ----------------
// A workgroup of NW waves owns TR = 4 NW output rows x 256 output columns of one plane.  Load phase: the TR + 3 input rows are dealt round-robin
// to the waves; a wave issues ALL its row loads (16 B per lane + the 3-column right halo by lanes 0..2) back to back, undoes the edge clamp,
// and parks the rows -- zero padding included -- in LDS.  One barrier.  Compute phase: wave w produces output rows 4w .. 4w+3 from LDS rows
// 4w .. 4w+6 (two aligned ds_read_b128 per lane and row: own four columns + the next four), same fmaf order as every other variant.
// Short-lived workgroups whose requests all leave at t = 0, like the float4 copy that reaches 6.2 TB/s; over-fetch (TR + 3) / TR.
template <int NW, int NT>
__global__ __launch_bounds__(64 * NW) void k_tile(P p, int colgroups, int tiles_y) {
    constexpr int TR = 4 * NW, ROWS = TR + 3, PITCH = 264;     // floats per LDS row (256 + 4 halo + 4 pad; 16-byte aligned rows)
    __shared__ float lds[ROWS * PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ty = blockIdx.x % tiles_y;
    const int cg = (blockIdx.x / tiles_y) % colgroups;
    const int pl = blockIdx.x / (tiles_y * colgroups);
    float ff[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) ff[a][b] = p.f[(3 - a) * 4 + (3 - b)];
    const float* xp = p.x + (size_t)pl * p.in_h * p.in_w;
    float* yp = p.y + (size_t)pl * p.out_h * p.out_w;
    const int ox = cg * 256 + lane * 4;
    const int ix0 = ox - p.pad;
    const int base = min(max(ix0, 0), p.in_w - 4);
    const int sh = base - ix0;
    const bool lane_dead = (ix0 >= p.in_w) || (ix0 + 3 < 0);
    const int ixh = cg * 256 + 256 - p.pad + lane;
    const bool halo_ok = lane < 3 && ixh >= 0 && ixh < p.in_w;
    const int ixh_c = min(max(ixh, 0), p.in_w - 1);
    const int oy0 = ty * TR;
    const int iy0 = oy0 - p.pad;
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int RPW = (ROWS + NW - 1) / NW;        // rows per wave in the load phase
    f4v m[RPW]; float h[RPW];
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wave + k * NW;                  // tile row
        const int iy = min(max(iy0 + r, 0), p.in_h - 1);
        const float* row = xp + (size_t)iy * p.in_w;
        if (r < ROWS) {
            m[k] = NT ? __builtin_nontemporal_load((const f4v*)(row + base)) : *(const f4v*)(row + base);
            h[k] = lane < 3 ? row[ixh_c] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wave + k * NW;
        if (r >= ROWS) continue;
        const int iy = iy0 + r;
        const bool row_ok = iy >= 0 && iy < p.in_h;
        f4v o;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = m[k][i];
#pragma unroll
            for (int d = 1; d <= 3; d++) {
                if (i - d >= 0) v = (sh == d) ? m[k][i - d] : v; else v = (sh == d) ? 0.f : v;
                if (i + d < 4) v = (sh == -d) ? m[k][i + d] : v; else v = (sh == -d) ? 0.f : v;
            }
            o[i] = (row_ok && !lane_dead) ? v : 0.f;
        }
        *(f4v*)(lds + r * PITCH + lane * 4) = o;
        if (lane < 4) lds[r * PITCH + 256 + lane] = (row_ok && halo_ok) ? h[k] : 0.f;    // lane 3 writes the pad word (zero)
    }
    __syncthreads();
    float win[7][8];
#pragma unroll
    for (int r = 0; r < 7; r++) {
        const f4v a = *(const f4v*)(lds + (4 * wave + r) * PITCH + lane * 4);
        const f4v b = *(const f4v*)(lds + (4 * wave + r) * PITCH + lane * 4 + 4);
#pragma unroll
        for (int i = 0; i < 4; i++) { win[r][i] = a[i]; win[r][4 + i] = b[i]; }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int oy = oy0 + 4 * wave + k;
        if (oy >= p.out_h) break;
        f4v sv;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc = __builtin_fmaf(win[k + j][v + i], ff[j][i], acc);
            sv[v] = acc * p.gain;
        }
        if (ox < p.out_w) __builtin_nontemporal_store(sv, (f4v*)(yp + (size_t)oy * p.out_w + ox));
    }
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 32;
    int C = 64, IH = argc > 2 ? atoi(argv[2]) : 257, IW = IH, pad = argc > 3 ? atoi(argv[3]) : 1;
    int OW = IW + 2 * pad - 3, OH = IH + 2 * pad - 3;
    int planes = N * C;
    size_t nx = (size_t)planes * IH * IW, ny = (size_t)planes * OH * OW;
    printf("FIR %dx%d -> %dx%d, planes %d, pad %d, bytes %.3f GB\n", IH, IW, OH, OW, planes, pad, (nx + ny) * 4 / 1e9);
    const int NBUF = 3;
    float *x[NBUF], *y, *yref, *f;
    for (int i = 0; i < NBUF; i++) CK(hipMalloc(&x[i], nx * 4 + 64));
    CK(hipMalloc(&y, ny * 4 + 64)); CK(hipMalloc(&yref, ny * 4 + 64)); CK(hipMalloc(&f, 64));
    std::vector<float> hx(nx), hf(16);
    unsigned s = 12345;
    for (size_t i = 0; i < nx; i++) { s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    float t1[4] = {1, 3, 3, 1};
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) hf[a * 4 + b] = t1[a] * t1[b] / 64.f;
    for (int i = 0; i < NBUF; i++) CK(hipMemcpy(x[i], hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(f, hf.data(), 64, hipMemcpyHostToDevice));
    P p{x[0], f, yref, IW, IH, OW, OH, planes, pad, 4.0f, 32};
    hipLaunchKernelGGL(k_naive, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> href(ny), hy(ny);
    CK(hipMemcpy(href.data(), yref, ny * 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto bench = [&](const char* name, auto launch, bool check) {
        CK(hipMemset(y, 0xff, ny * 4));
        for (int w = 0; w < 2; w++) launch(x[w % NBUF], y);
        CK(hipDeviceSynchronize());
        if (check) {
            CK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < ny; i++) if (!(hy[i] == href[i])) bad++;
            if (bad) {
                printf("  !! %s: %zu mismatches\n", name, bad);
                int shown = 0;
                for (size_t i = 0; i < ny && shown < 24; i++) if (!(hy[i] == href[i])) { printf("     plane %zu row %zu col %zu: got %g want %g\n", i / ((size_t)OH * OW), (i / OW) % OH, i % OW, hy[i], href[i]); shown++; }
            }
        }
        float best = 1e9, tot = 0; int reps = 12;
        for (int r = 0; r < reps; r++) {
            // four back-to-back launches per bracket: the host-side cost of a launch (the C ABI plans inside its call) hides behind the previous kernel
            CK(hipEventRecord(e0)); for (int q = 0; q < 4; q++) launch(x[(r + q) % NBUF], y); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms *= 0.25f; best = ms < best ? ms : best; tot += ms;
        }
        double gb = (nx + ny) * 4 / 1e9;
        printf("%-34s avg %8.4f ms  min %8.4f ms   %7.1f GB/s (min %7.1f)  %5.1f%% of 8 TB/s\n", name, tot / reps, best, gb / (tot / reps) * 1e3, gb / best * 1e3, gb / (tot / reps) * 1e3 / 80.0);
    };
    bench("copy float4 (same bytes)", [&](const float* xi, float* yo) {
        long n4 = (long)((nx < ny ? nx : ny) / 4); hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const float4*)xi, (float4*)yo, n4); }, false);
    // the production library through its C ABI, same harness
    const char* so_path = getenv("SGV_LIB") ? getenv("SGV_LIB") : "stylegan-v_amd/csrc/libsgv_hip.so";
    void* so = dlopen(so_path, RTLD_NOW);
    typedef int (*ufd_fn)(const sgv_upfirdn2d_params*, int, void*);
    ufd_fn sgv = so ? (ufd_fn)dlsym(so, "sgv_upfirdn2d") : nullptr;
    if (sgv) {
        sgv_upfirdn2d_params q{};
        q.f = f; q.up_x = q.up_y = q.down_x = q.down_y = 1; q.pad_x0 = q.pad_x1 = q.pad_y0 = q.pad_y1 = pad; q.flip = 0; q.gain = 4.0f;
        q.in_w = IW; q.in_h = IH; q.in_c = C; q.in_n = N; q.in_sw = 1; q.in_sh = IW; q.in_sc = (int64_t)IW * IH; q.in_sn = (int64_t)IW * IH * C;
        q.f_w = q.f_h = 4; q.f_sw = 1; q.f_sh = 4; q.out_w = OW; q.out_h = OH; q.out_sw = 1; q.out_sh = OW; q.out_sc = (int64_t)OW * OH; q.out_sn = (int64_t)OW * OH * C;
        bench("libsgv_hip sgv_upfirdn2d (C ABI)", [&](const float* xi, float* yo) { sgv_upfirdn2d_params r = q; r.x = xi; r.y = yo; if (sgv(&r, 0, nullptr)) printf("sgv error\n"); }, true);
    } else printf("libsgv_hip.so not found: %s\n", dlerror());
#define RUNC(NTL, NTS, U) { std::string nm = std::string("copy2 ntl") + #NTL + " nts" + #NTS + " U" + #U; \
      bench(nm.c_str(), [&](const float* xi, float* yo) { long n4 = (long)((nx < ny ? nx : ny) / 4); hipLaunchKernelGGL((k_copy2<NTL, NTS, U>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, (const fv4*)xi, (fv4*)yo, n4); }, false); }
    RUNC(0, 0, 1) RUNC(1, 1, 4)
    for (int sh : {8, 16}) {
        int strips4 = (OH + sh - 1) / sh; int cgs4 = (OW + 255) / 256; long waves4 = (long)planes * cgs4 * strips4; P q4 = p; q4.strip_h = sh;
#define RUN4(PFV, NTV) { std::string nm = std::string("V3 cols dwordx4 PF") + #PFV + " NT" + #NTV + " strip " + std::to_string(sh); \
          bench(nm.c_str(), [&](const float* xi, float* yo) { P r = q4; r.x = xi; r.y = yo; hipLaunchKernelGGL((k_cols4<PFV, NTV>), dim3((unsigned)((waves4 + 3) / 4)), dim3(256), 0, 0, r, cgs4, strips4); }, true); }
        RUN4(1, 1) RUN4(2, 1) RUN4(4, 1)
        if (OW % 4 == 0 && OH % sh == 0) {   // (and sh % 8 == 0)
#define RUN5(PFV) { std::string nm = std::string("V4 asm loads, counted waits PF") + #PFV + " strip " + std::to_string(sh); \
          bench(nm.c_str(), [&](const float* xi, float* yo) { P r = q4; r.x = xi; r.y = yo; hipLaunchKernelGGL((k_cols4_asm<PFV>), dim3((unsigned)((waves4 + 3) / 4)), dim3(256), 0, 0, r, cgs4, strips4, getenv("UFD_DRAIN") ? 1 : 0); }, true); }
        RUN5(1) RUN5(2) RUN5(4) if (sh % 16 == 0) RUN5(8)
        }
#define RUN4A(PFV, NTV, AB) { std::string nm = std::string("V3 ablation ") + #AB + " PF" + #PFV + " NT" + #NTV + " strip " + std::to_string(sh); \
          bench(nm.c_str(), [&](const float* xi, float* yo) { P r = q4; r.x = xi; r.y = yo; hipLaunchKernelGGL((k_cols4<PFV, NTV, AB>), dim3((unsigned)((waves4 + 3) / 4)), dim3(256), 0, 0, r, cgs4, strips4); }, false); }
        RUN4A(4, 1, 1) RUN4A(4, 1, 2) RUN4A(1, 1, 2) RUN4A(4, 2, 2)
    }

    {   // V7: the PRODUCT's upfirdn2d_tile_kernel launched directly from this harness (same parameters as the C ABI builds)
        tile_params tp{};
        tp.f = f; tp.flip = 0; tp.gain = 4.0f; tp.in_w = IW; tp.in_h = IH; tp.out_w = OW; tp.out_h = OH; tp.planes = planes; tp.f_w = tp.f_h = 4; tp.f_sw = 1; tp.f_sh = 4;
        tp.pad_x = tp.pad_y = pad; tp.lpr_log2 = 6; tp.col_groups = (OW / 4 + 63) / 64; tp.row_tiles = (OH + 15) / 16; tp.nt_store = 1;
        tp.ep_act = 1; tp.ep_gain = 1.f; tp.ep_clamp = -1.f; tp.chans = C;
        const long blocks7 = (long)planes * tp.col_groups * tp.row_tiles;
        const size_t lds7 = (size_t)tile_lds_floats(6) * 4;
        if (OW % 4 == 0) {
            bench("V7 product tile kernel, direct launch", [&](const float* xi, float* yo) { tile_params r = tp; r.x = xi; r.y = yo;
                hipLaunchKernelGGL((upfirdn2d_tile_kernel<float, 0, 0, true, true, true>), dim3((unsigned)blocks7), dim3(256), lds7, 0, r); }, true);
            bench("V7b product tile kernel (run-time pitches), direct", [&](const float* xi, float* yo) { tile_params r = tp; r.x = xi; r.y = yo;
                hipLaunchKernelGGL((upfirdn2d_tile_kernel<float, 0, 0, false, true, true>), dim3((unsigned)blocks7), dim3(256), lds7, 0, r); }, true);
        } else if (OW % 4 == 1) {
            bench("V7 product tile kernel XTRA, direct launch", [&](const float* xi, float* yo) { tile_params r = tp; r.x = xi; r.y = yo;
                hipLaunchKernelGGL((upfirdn2d_tile_kernel<float, 1, 0, true, true, true>), dim3((unsigned)blocks7), dim3(256), lds7, 0, r); }, true);
        }
    }
    if (OW % 4 == 0) {
#define RUN6(NWV, NTV) { const int TRv = 4 * NWV; int tiles_y = (OH + TRv - 1) / TRv; int cgs6 = (OW + 255) / 256; long blocks = (long)planes * cgs6 * tiles_y; \
          std::string nm = std::string("V6 LDS tile, loads up front, ") + std::to_string(TRv) + " rows NT" + #NTV; \
          bench(nm.c_str(), [&](const float* xi, float* yo) { P r = p; r.x = xi; r.y = yo; hipLaunchKernelGGL((k_tile<NWV, NTV>), dim3((unsigned)blocks), dim3(64 * NWV), 0, 0, r, cgs6, tiles_y); }, true); }
        RUN6(4, 0) RUN6(4, 1) RUN6(8, 0) RUN6(8, 1) RUN6(16, 0)
    }
    for (int sh : std::vector<int>{}) {
        int strips = (OH + sh - 1) / sh;
        { int cgs = (OW + 63) / 64; long waves = (long)planes * cgs * strips; P q = p; q.strip_h = sh; q.y = y;
          std::string nm = "V1 cols dword strip " + std::to_string(sh);
          bench(nm.c_str(), [&](const float* xi, float* yo) { P r = q; r.x = xi; r.y = yo; hipLaunchKernelGGL(k_cols<0>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, r, cgs, strips); }, true); }
        { int cgs = (OW + 127) / 128; long waves = (long)planes * cgs * strips; P q = p; q.strip_h = sh; q.y = y;
          std::string nm = "V2 cols dwordx2 strip " + std::to_string(sh);
          bench(nm.c_str(), [&](const float* xi, float* yo) { P r = q; r.x = xi; r.y = yo; hipLaunchKernelGGL(k_cols2, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, r, cgs, strips); }, true); }
    }
    return 0;
}
