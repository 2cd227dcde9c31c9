// Round-6 kernel laboratory for the upfirdn2d FIR passes (up = down = 1, 4x4 taps): what separates the LDS-tile kernel from the float4 copy.
// Standalone (no torch): hipcc --offload-arch=gfx950 -O3 tools/ufd_lab6.hip -o tools/ufd_lab6 && tools/ufd_lab6 [N] [IH] [pad]
// Every variant is checked bit-for-bit against a naive per-output kernel and timed two ways: one launch per HIP-event bracket (what the library's per-launch
// profiler and tools/ops_bench.py see) and four launches per bracket (steady state; what a kernel trace reports).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <string>
#include <string.h>
#include <functional>
#include <algorithm>
#include "../include/sgv_ops.h"
#include "../stylegan-v_amd/csrc/sgv_runtime.hip"
#include "../stylegan-v_amd/csrc/upfirdn2d.hip"
#pragma clang fp contract(off)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct P { const float* x; const float* f; float* y; int in_w, in_h, out_w, out_h, planes, pad; float gain; unsigned magic; };
typedef float fv4 __attribute__((ext_vector_type(4)));
typedef float fv4u __attribute__((ext_vector_type(4), aligned(4)));

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

// ---------------- copies: the ceiling, and what misalignment of the 16-byte requests costs ----------------
// SO / DO: source / destination displaced by that many floats from 16-byte alignment
template <int NTL, int NTS, int U, int SO, int DO>
__global__ __launch_bounds__(256) void k_copy2(const float* __restrict__ a, float* __restrict__ b, long n4) {
    long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
    fv4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { long i = base + u * 256; if (i < n4) v[u] = NTL ? __builtin_nontemporal_load((const fv4u*)(a + SO + 4 * i)) : *(const fv4u*)(a + SO + 4 * i); }
#pragma unroll
    for (int u = 0; u < U; u++) { long i = base + u * 256; if (i < n4) { if (NTS) __builtin_nontemporal_store(v[u], (fv4u*)(b + DO + 4 * i)); else *(fv4u*)(b + DO + 4 * i) = v[u]; } }
}

// row copy: a workgroup moves 16 rows x 256 columns of one plane from a pitch-in_w image to a pitch-out_w image (the FIR's request pattern without its halo, LDS or arithmetic)
template <int NT>
__global__ __launch_bounds__(256) void k_rowcopy(P p, int tiles_y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ty = blockIdx.x % tiles_y, pl = blockIdx.x / tiles_y;
    const float* xp = p.x + (size_t)pl * p.in_h * p.in_w;
    float* yp = p.y + (size_t)pl * p.out_h * p.out_w;
    fv4 m[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int r = ty * 16 + wave + 4 * k; m[k] = *(const fv4u*)(xp + (size_t)min(r, p.in_h - 1) * p.in_w + 4 * lane); }
#pragma unroll
    for (int k = 0; k < 4; k++) { const int r = ty * 16 + wave + 4 * k; if (r < p.out_h) { if (NT) __builtin_nontemporal_store(m[k], (fv4u*)(yp + (size_t)r * p.out_w + 4 * lane)); else *(fv4u*)(yp + (size_t)r * p.out_w + 4 * lane) = m[k]; } }
}

// ---------------- V6 (round 3): LDS tile, every load of the workgroup up front; XCD = 1: the 16 row tiles of a plane run on ONE XCD (blockIdx % 8) ----------------
template <int NW, int NT, int XCD>
__global__ __launch_bounds__(64 * NW) void k_tile(P p, int colgroups, int tiles_y) {
    constexpr int TR = 4 * NW, ROWS = TR + 3, PITCH = 264;
    __shared__ float lds[ROWS * PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ty, cg, pl;
    if (XCD) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; ty = j % tiles_y; cg = 0; pl = (j / tiles_y) * 8 + xcd; }
    else { ty = blockIdx.x % tiles_y; cg = (blockIdx.x / tiles_y) % colgroups; pl = blockIdx.x / (tiles_y * colgroups); }
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
    constexpr int RPW = (ROWS + NW - 1) / NW;
    fv4 m[RPW]; float h[RPW];
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wave + k * NW;
        const int iy = min(max(iy0 + r, 0), p.in_h - 1);
        const float* row = xp + (size_t)iy * p.in_w;
        if (r < ROWS) {
            m[k] = NT ? __builtin_nontemporal_load((const fv4u*)(row + base)) : *(const fv4u*)(row + base);
            h[k] = lane < 3 ? row[ixh_c] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wave + k * NW;
        if (r >= ROWS) continue;
        const int iy = iy0 + r;
        const bool row_ok = iy >= 0 && iy < p.in_h;
        fv4 o;
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
        *(fv4*)(lds + r * PITCH + lane * 4) = o;
        if (lane < 4) lds[r * PITCH + 256 + lane] = (row_ok && halo_ok) ? h[k] : 0.f;
    }
    __syncthreads();
    float win[7][8];
#pragma unroll
    for (int r = 0; r < 7; r++) {
        const fv4 a = *(const fv4*)(lds + (4 * wave + r) * PITCH + lane * 4);
        const fv4 b = *(const fv4*)(lds + (4 * wave + r) * PITCH + lane * 4 + 4);
#pragma unroll
        for (int i = 0; i < 4; i++) { win[r][i] = a[i]; win[r][4 + i] = b[i]; }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int oy = oy0 + 4 * wave + k;
        if (oy >= p.out_h) break;
        fv4 sv;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc = __builtin_fmaf(win[k + j][v + i], ff[j][i], acc);
            sv[v] = acc * p.gain;
        }
        if (ox < p.out_w) __builtin_nontemporal_store(sv, (fv4*)(yp + (size_t)oy * p.out_w + ox));
    }
}

// ---------------- V8: the tile's input rows as ONE linear span ----------------
// A tile spans the whole row (in_w <= 260), so its 19 input rows are one contiguous run of the plane: it is fetched with naturally aligned 16-byte
// requests (start aligned DOWN, 256 threads x 5 requests, all at t = 0 -- the copy's request pattern, whatever in_w mod 4 and the padding are) and scattered
// into the padded LDS image (row pitch 264, position q = column q - PAD, zero padding written by the workgroup).  Compute phase as V6.
// XTRA (out_w = 4 k + 1): lane 63 produces a fifth column.  STAGE = 1: the 16 output rows (one contiguous run of the output plane) are parked in LDS and
// leave as naturally aligned 16-byte stores instead of rows of pitch out_w.
template <int PAD, int XTRA, int XCD, int STAGE, int NTL>
__global__ __launch_bounds__(256) void k_span(P p, int tiles_y) {
    constexpr int TR = 16, ROWS = 19, PITCH = 264, NOUT = 4 + XTRA, NL = 5;
    __shared__ __attribute__((aligned(16))) float lds[ROWS * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ty, pl;
    if (XCD) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; ty = j % tiles_y; pl = (j / tiles_y) * 8 + xcd; }
    else { ty = blockIdx.x % tiles_y; pl = blockIdx.x / tiles_y; }
    float ff[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) ff[a][b] = p.f[(3 - a) * 4 + (3 - b)];
    const int oy0 = ty * TR, iy0 = oy0 - PAD;
    const int ra = max(iy0, 0), rb = min(iy0 + ROWS, p.in_h);
    const int span = (rb - ra) * p.in_w;
    const float* src = p.x + (size_t)pl * p.in_h * p.in_w + (size_t)ra * p.in_w;
    const int a = (int)(((uintptr_t)src >> 2) & 3);
    const fv4* src4 = (const fv4*)(src - a);
    const int n4 = (span + a + 3) >> 2;
    fv4 m[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int idx = tid + 256 * k;
        m[k] = fv4{0.f, 0.f, 0.f, 0.f};
        if (idx < n4) m[k] = NTL ? __builtin_nontemporal_load(src4 + idx) : src4[idx];
    }
    // zero padding while the loads are in flight: the columns left / right of every row, and whole rows above / below the plane
    const int nz = PITCH - p.in_w;     // <= 16
    for (int i = tid; i < ROWS * 16; i += 256) {
        const int r = i >> 4, j = i & 15;
        if (j < nz) lds[r * PITCH + (j < PAD ? j : p.in_w + j)] = 0.f;
    }
    for (int r = 0; r < ROWS; r++) {
        const int iy = iy0 + r;
        if (iy < 0 || iy >= p.in_h) for (int q = tid; q < PITCH; q += 256) lds[r * PITCH + q] = 0.f;
    }
    const int lrow0 = (ra - iy0) * PITCH + PAD;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int idx = tid + 256 * k;
        if (idx < n4) {
            const int t0 = 4 * idx - a;
            const int tt = max(t0, 0);
            const int rr = (int)__umulhi((unsigned)tt, p.magic);
            const int cc = tt - rr * p.in_w;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = t0 + i;
                int c = cc + (t - tt), r = rr;
                if (c >= p.in_w) { c -= p.in_w; r++; }
                if (t >= 0 && t < span) lds[lrow0 + r * PITCH + c] = m[k][i];
            }
        }
    }
    __syncthreads();
    float win[7][8];
#pragma unroll
    for (int r = 0; r < 7; r++) {
        const fv4 a0 = *(const fv4*)(lds + (4 * wave + r) * PITCH + lane * 4);
        const fv4 b0 = *(const fv4*)(lds + (4 * wave + r) * PITCH + lane * 4 + 4);
#pragma unroll
        for (int i = 0; i < 4; i++) { win[r][i] = a0[i]; win[r][4 + i] = b0[i]; }
    }
    float* yp = p.y + (size_t)pl * p.out_h * p.out_w;
    const int ox = lane * 4;
    float o[4][NOUT];
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int v = 0; v < NOUT; v++) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc = __builtin_fmaf(win[k + j][v + i], ff[j][i], acc);
            o[k][v] = acc * p.gain;
        }
    }
    if constexpr (!STAGE) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int oy = oy0 + 4 * wave + k;
            if (oy >= p.out_h) break;
            float* yr = yp + (size_t)oy * p.out_w + ox;
            if (ox + 4 <= p.out_w) __builtin_nontemporal_store(fv4{o[k][0], o[k][1], o[k][2], o[k][3]}, (fv4u*)yr);
            if constexpr (XTRA) { if (ox + 4 == p.out_w - 1) yr[4] = o[k][4]; }
        }
    } else {
        const int nrows = min(TR, p.out_h - oy0);
        const int ospan = nrows * p.out_w;
        float* dst = yp + (size_t)oy0 * p.out_w;
        const int a2 = (int)(((uintptr_t)dst >> 2) & 3);
        __syncthreads();     // every wave has its window in registers: the LDS image can go
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int q = a2 + (4 * wave + k) * p.out_w + ox;
#pragma unroll
            for (int v = 0; v < 4; v++) if (ox + v < p.out_w) lds[q + v] = o[k][v];
            if constexpr (XTRA) { if (ox + 4 == p.out_w - 1) lds[q + 4] = o[k][4]; }
        }
        __syncthreads();
        const int n4o = (ospan + a2 + 3) >> 2;
        for (int idx = tid; idx < n4o; idx += 256) {
            const fv4 v = *(const fv4*)(lds + 4 * idx);
            const int t0 = 4 * idx - a2;
            if (t0 >= 0 && t0 + 4 <= ospan) __builtin_nontemporal_store(v, (fv4*)(dst + t0));
            else {
#pragma unroll
                for (int i = 0; i < 4; i++) if (t0 + i >= 0 && t0 + i < ospan) dst[t0 + i] = v[i];
            }
        }
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
    for (int i = 0; i < NBUF; i++) CK(hipMalloc(&x[i], nx * 4 + 256));
    CK(hipMalloc(&y, ny * 4 + 256)); CK(hipMalloc(&yref, ny * 4 + 256)); CK(hipMalloc(&f, 64));
    std::vector<float> hx(nx), hf(16);
    unsigned s = 12345;
    for (size_t i = 0; i < nx; i++) { s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    float t1[4] = {1, 3, 3, 1};
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) hf[a * 4 + b] = t1[a] * t1[b] / 64.f;
    for (int i = 0; i < NBUF; i++) CK(hipMemcpy(x[i], hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(f, hf.data(), 64, hipMemcpyHostToDevice));
    P p{x[0], f, yref, IW, IH, OW, OH, planes, pad, 4.0f, (unsigned)(0x100000000ull / (unsigned)IW + 1)};
    hipLaunchKernelGGL(k_naive, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> href(ny), hy(ny);
    CK(hipMemcpy(href.data(), yref, ny * 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = (nx + ny) * 4 / 1e9;
    const char* only = getenv("UFD_ONLY");
    // Protocol (profiles/r06_c3: the first ~30 launches after an idle period run up to 40 % slow -- the chip's own transient, every kernel shows it): a variant is
    // checked once, then every round gives it WARM untimed launches followed by REPS launches inside one event bracket; the rounds cycle through all variants
    // (A B C A B C ...) and the table reports the median round.
    struct variant { std::string name; std::function<void(const float*, float*)> launch; std::vector<float> ms; };
    std::vector<variant> variants;
    auto bench = [&](const char* name, auto launch, bool check) {
        if (only && !strstr(name, only) && !(getenv("UFD_ONLY2") && strstr(name, getenv("UFD_ONLY2")))) return;
        if (check) {
            CK(hipMemset(y, 0xff, ny * 4));
            launch(x[0], y);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < ny; i++) if (__builtin_memcmp(&hy[i], &href[i], 4) != 0) bad++;
            if (bad) {
                printf("  !! %s: %zu mismatches\n", name, bad);
                int shown = 0;
                for (size_t i = 0; i < ny && shown < 8; i++) if (__builtin_memcmp(&hy[i], &href[i], 4) != 0) { printf("     plane %zu row %zu col %zu: got %g want %g\n", i / ((size_t)OH * OW), (i / OW) % OH, i % OW, hy[i], href[i]); shown++; }
            }
        }
        variants.push_back(variant{name, launch, {}});
    };
    auto run_all = [&]() {
        const int rounds = getenv("UFD_ROUNDS") ? atoi(getenv("UFD_ROUNDS")) : 5, WARM = 24, REPS = 24;
        for (int w = 0; w < 60; w++) variants[0].launch(x[w % NBUF], y);     // leave the idle state behind
        for (int rd = 0; rd < rounds; rd++)
            for (auto& v : variants) {
                for (int w = 0; w < WARM; w++) v.launch(x[w % NBUF], y);
                CK(hipEventRecord(e0)); for (int q = 0; q < REPS; q++) v.launch(x[q % NBUF], y); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / REPS);
            }
        for (auto& v : variants) {
            std::vector<float> t = v.ms; std::sort(t.begin(), t.end());
            const float med = t[t.size() / 2];
            printf("%-46s median %7.4f ms (min %7.4f max %7.4f) %7.1f GB/s %5.1f%% of 8 TB/s\n", v.name.c_str(), med, t.front(), t.back(), gb / med * 1e3, gb / med * 1e3 / 80.0);
        }
        fflush(stdout);
    };
    const long n4c = (long)((nx < ny ? nx : ny) / 4) - 1;
#define RUNC(NTL, NTS, U, SO, DO) { std::string nm = std::string("copy ntl") + #NTL + " nts" + #NTS + " U" + #U + " src+" + #SO + " dst+" + #DO; \
      bench(nm.c_str(), [=](const float* xi, float* yo) { hipLaunchKernelGGL((k_copy2<NTL, NTS, U, SO, DO>), dim3((unsigned)((n4c + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, xi, yo, n4c); }, false); }
    RUNC(1, 1, 4, 0, 0) RUNC(0, 1, 4, 0, 0) RUNC(0, 1, 4, 1, 0) RUNC(0, 1, 4, 0, 1) RUNC(0, 1, 4, 1, 1) RUNC(0, 0, 4, 0, 0)
    const int tiles16 = (OH + 15) / 16;
    bench("rowcopy 16 rows x 256, pitch in_w -> out_w NT", [=](const float* xi, float* yo) { P r = p; r.x = xi; r.y = yo; hipLaunchKernelGGL((k_rowcopy<1>), dim3((unsigned)(planes * tiles16)), dim3(256), 0, 0, r, tiles16); }, false);

    for (int xc = 0; xc < 2; xc++) {   // the product's tile kernel, launched directly
        tile_params tp{};
        tp.f = f; tp.flip = 0; tp.gain = 4.0f; tp.in_w = IW; tp.in_h = IH; tp.out_w = OW; tp.out_h = OH; tp.planes = planes; tp.f_w = tp.f_h = 4; tp.f_sw = 1; tp.f_sh = 4;
        tp.pad_x = tp.pad_y = pad; tp.lpr_log2 = 6; tp.col_groups = (OW / 4 + 63) / 64; tp.row_tiles = (OH + 15) / 16; tp.nt_store = 1;
        tp.ep_act = 1; tp.ep_gain = 1.f; tp.ep_clamp = -1.f; tp.chans = C;
        tp.xcd_blocks = xc ? (planes * tp.col_groups / 8) * 8 * tp.row_tiles : 0;
        const long blocks7 = (long)planes * tp.col_groups * tp.row_tiles;
        tp.lds_amax_word = tile_lds_floats(6);
        const size_t lds7 = (size_t)(tile_lds_floats(6) + 4) * 4;
        std::string nm = std::string("V7 product tile kernel LAB_OFF=") + std::to_string(SGV_TILE_LAB_OFF) + (xc ? " xcd1" : " xcd0");
        if (OW % 4 == 0)
            bench(nm.c_str(), [=](const float* xi, float* yo) { tile_params r = tp; r.x = xi; r.y = yo;
                hipLaunchKernelGGL((upfirdn2d_tile_kernel<float, 0, 0, true, true, true>), dim3((unsigned)blocks7), dim3(256), lds7, 0, r); }, true);
        else
            bench(nm.c_str(), [=](const float* xi, float* yo) { tile_params r = tp; r.x = xi; r.y = yo;
                hipLaunchKernelGGL((upfirdn2d_tile_kernel<float, 1, 0, true, true, true>), dim3((unsigned)blocks7), dim3(256), lds7, 0, r); }, true);
    }
    if (getenv("UFD_PRODUCT_ONLY")) { run_all(); return 0; }
    if (OW % 4 == 0) {
#define RUN6(NWV, NTV, XC) { const int TRv = 4 * NWV; int tiles_y = (OH + TRv - 1) / TRv; long blocks = (long)planes * tiles_y; \
          std::string nm = std::string("V6 LDS tile ") + std::to_string(TRv) + " rows NTload" + #NTV + " xcd" + #XC; \
          bench(nm.c_str(), [=](const float* xi, float* yo) { P r = p; r.x = xi; r.y = yo; hipLaunchKernelGGL((k_tile<NWV, NTV, XC>), dim3((unsigned)blocks), dim3(64 * NWV), 0, 0, r, 1, tiles_y); }, true); }
        RUN6(4, 0, 0) RUN6(4, 0, 1) RUN6(4, 1, 0) RUN6(4, 1, 1)
    }
#define RUN8(PADV, XT, XC, ST, NTL) if (pad == PADV && (OW % 4 == 1) == (XT == 1)) { \
          std::string nm = std::string("V8 span pad") + #PADV + " xtra" + #XT + " xcd" + #XC + " stage" + #ST + " ntl" + #NTL; \
          bench(nm.c_str(), [=](const float* xi, float* yo) { P r = p; r.x = xi; r.y = yo; hipLaunchKernelGGL((k_span<PADV, XT, XC, ST, NTL>), dim3((unsigned)(planes * tiles16)), dim3(256), 0, 0, r, tiles16); }, true); }
    RUN8(1, 0, 0, 0, 0) RUN8(1, 0, 1, 0, 0) RUN8(1, 0, 0, 0, 1) RUN8(1, 0, 1, 0, 1)
    RUN8(2, 1, 0, 0, 0) RUN8(2, 1, 1, 0, 0) RUN8(2, 1, 0, 1, 0) RUN8(2, 1, 1, 1, 0) RUN8(2, 1, 1, 1, 1)
    run_all();
    return 0;
}
