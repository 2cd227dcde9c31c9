// upfirdn2d for gfx950: pad -> zero-insert -> FIR -> decimate.
//
// Semantics follow the reference op exactly (src/torch_utils/ops/upfirdn2d.cu:43-48,60-89 for
// the receptive-field / tap walk, upfirdn2d.cpp:32-33 for the output size).  Three kernels (the third,
// upfirdn2d_lanes_kernel, is described at its definition):
//
//  * upfirdn2d_generic_kernel  -- one lane per output element, any strides / factors / filter
//    size / dtype (incl. fp64, channels_last).  The correctness backstop.
//  * upfirdn2d_rows_kernel     -- the hot path.  Contiguous NCHW, up/down in {1,2} per axis (never
//    both), filter <= 4x4 (2-D) or a 1-D pass of <= 12 taps.  HBM-bound streaming design with NO LDS and NO barriers:
//      - a wave owns a strip of output rows of one (or, for narrow images, several) planes;
//        each lane owns VEC=4 adjacent output columns (16 B of fp32 per row -> one dwordx4 store,
//        the wave writes 1 KiB contiguous per row);
//      - the lane keeps a sliding window of the input rows its outputs need in VGPRs
//        (WR x NEED floats) and walks down the strip: every input element is loaded from global
//        memory once per strip (vertical halo = (FH-1)/strip rows, absorbed by L2 because the four
//        waves of a workgroup own four consecutive strips of the same plane), horizontal halo is
//        the 3..6 extra columns each lane loads (L1/TA hits on the neighbour lane's lines);
//      - filter taps are wave-uniform -> the compiler keeps them in SGPRs (s_load), the MAC loop is
//        pure v_fma_f32 with an SGPR operand;
//      - up/down factors, padded filter size and the launch-constant phase (pad mod up) are
//        template parameters, so every window index is static and lives in a register.
//    Tap order per output is ascending input row, then ascending input column, as one fmaf chain:
//    the same order as the reference loop, so results are bit-identical to the scalar oracle
//    (oracle/sgv_oracle.c) -- zero-padded taps contribute fma(0,f,acc)==acc.
//
// Algorithmic bytes per launch (what the roofline is computed from):
//    (numel(x) + numel(y)) * sizeof(T)   [+ 4*fw*fh, negligible]       

#include "sgv_common.h"
#include <type_traits>
#include <stdlib.h>

#pragma clang fp contract(off)

namespace {

__host__ __device__ __forceinline__ int floor_div(int a, int b) {
    // floor(a / b) for b > 0 without relying on negative-division rounding (upfirdn2d.cu:20-24).
    int t = 1 - a / b;
    return (a + t * b) / b - t;
}

template <typename A> __device__ __forceinline__ A fma_acc(A a, A b, A c);
template <> __device__ __forceinline__ float fma_acc<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_acc<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---------------------------------------------------------------------------------------------
// Generic kernel.

struct generic_params {
    const void* x;
    const float* f;
    void* y;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0, flip;
    float gain;
    int in_w, in_h, in_c, in_n;
    int64_t in_sw, in_sh, in_sc, in_sn;
    int f_w, f_h;
    int64_t f_sw, f_sh;
    int out_w, out_h;
    int64_t out_sw, out_sh, out_sc, out_sn;
    int64_t total;  // out_w*out_h*in_c*in_n
    int chan_minor;  // 1: iterate channels fastest (channels_last), 0: x fastest
};

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_generic_kernel(generic_params p) {
    typedef typename sgv_traits<T>::acc_t acc_t;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < p.total; idx += (int64_t)gridDim.x * blockDim.x) {
        int ox, oy, c, n;
        int64_t r = idx;
        if (p.chan_minor) {
            c = (int)(r % p.in_c); r /= p.in_c;
            ox = (int)(r % p.out_w); r /= p.out_w;
            oy = (int)(r % p.out_h); n = (int)(r / p.out_h);
        } else {
            ox = (int)(r % p.out_w); r /= p.out_w;
            oy = (int)(r % p.out_h); r /= p.out_h;
            c = (int)(r % p.in_c); n = (int)(r / p.in_c);
        }
        // Receptive field (upfirdn2d.cu:43-48, 60-65).
        int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
        int in_y = min(max(floor_div(mid_y, p.up_y), 0), p.in_h);
        int h = min(max(floor_div(mid_y + p.f_h, p.up_y), 0), p.in_h) - in_y;
        int fy = mid_y + p.f_h - (in_y + 1) * p.up_y;
        int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
        int in_x = min(max(floor_div(mid_x, p.up_x), 0), p.in_w);
        int w = min(max(floor_div(mid_x + p.f_w, p.up_x), 0), p.in_w) - in_x;
        int fx = mid_x + p.f_w - (in_x + 1) * p.up_x;
        int step_x = -p.up_x, step_y = -p.up_y;
        if (p.flip) { fy = p.f_h - 1 - fy; fx = p.f_w - 1 - fx; step_x = p.up_x; step_y = p.up_y; }

        const T* xp = (const T*)p.x + in_x * p.in_sw + in_y * p.in_sh + c * p.in_sc + n * p.in_sn;
        acc_t v = 0;
        for (int j = 0; j < h; j++) {
            const float* fp = p.f + (fy + j * step_y) * p.f_sh + fx * p.f_sw;
            const T* xr = xp + j * p.in_sh;
            for (int i = 0; i < w; i++)
                v = fma_acc<acc_t>(sgv_traits<T>::load(xr + i * p.in_sw), (acc_t)fp[i * step_x * p.f_sw], v);
        }
        v *= (acc_t)p.gain;
        sgv_traits<T>::store((T*)p.y + ox * p.out_sw + oy * p.out_sh + c * p.out_sc + n * p.out_sn, v);
    }
}

// ---------------------------------------------------------------------------------------------
// Row-walker kernel.

struct rows_params {
    const void* x;
    const float* f;
    void* y;
    int pad_x0, pad_y0, flip;
    float gain;
    int in_w, in_h, out_w, out_h;
    int planes;  // n*c
    int f_w, f_h;
    int64_t f_sw, f_sh;
    int lpr_log2;        // lanes per image row = 1 << lpr_log2 (<= 64)
    int col_groups;      // ceil(column blocks / lanes per row)
    int strips;          // strips per plane
    int strip_h;         // output rows per strip (multiple of UPY)
    int plane_groups;    // ceil(planes / planes per wave)
    int xtra;            // 1: the lane owning columns [out_w-1-VEC, out_w-1) also writes column out_w-1
};

constexpr int VEC = 4;

__device__ __forceinline__ float dpp_wave_shl1(float v, float fill63) {  // lane i <- lane i+1, lane 63 <- fill63
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill63), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_wave_shr1(float v, float fill0) {  // lane i <- lane i-1, lane 0 <- fill0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill0), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

template <typename T, int N> struct vec_of;
template <int N> struct vec_of<float, N> { typedef float type __attribute__((ext_vector_type(N), aligned(4))); };
template <int N> struct vec_of<sgv_half_t, N> { typedef uint16_t type __attribute__((ext_vector_type(N), aligned(2))); };
template <int N> struct vec_of<sgv_bf16_t, N> { typedef uint16_t type __attribute__((ext_vector_type(N), aligned(2))); };

template <typename T> __device__ __forceinline__ float widen(uint16_t b);
template <> __device__ __forceinline__ float widen<sgv_half_t>(uint16_t b) { sgv_half_t h{b}; return sgv_traits<sgv_half_t>::load(&h); }
template <> __device__ __forceinline__ float widen<sgv_bf16_t>(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
template <typename T> __device__ __forceinline__ uint16_t narrow(float v) { T t; sgv_traits<T>::store(&t, v); return t.bits; }

// Load N contiguous elements starting at p (element-aligned only) into dst[0..N) as fp32.  NTL: non-temporal requests (data nobody re-reads: the loads of a
// streaming pass then leave the caches to the lines that ARE shared, and a plain float4 copy gains 7 % from them alone: profiles/r06_c1_ufd_lab6.log).
template <typename V, bool NTL> __device__ __forceinline__ V load_vec(const void* p) {
    if constexpr (NTL) return __builtin_nontemporal_load((const V*)p);
    else return *(const V*)p;
}
template <typename T, int N, bool NTL = false> struct row_loader {
    static __device__ __forceinline__ void run(const T* p, float* dst) {
        if constexpr (N >= 8 && sizeof(T) == 2) {     // eight 16-bit elements: one 16-byte request per lane (the tile kernel's CPL = 8 form)
            typename vec_of<T, 8>::type v = load_vec<typename vec_of<T, 8>::type, NTL>(p);
#pragma unroll
            for (int i = 0; i < 8; i++) dst[i] = widen<T>(v[i]);
            row_loader<T, N - 8, NTL>::run(p + 8, dst + 8);
        } else if constexpr (N >= 4) {
            typename vec_of<T, 4>::type v = load_vec<typename vec_of<T, 4>::type, NTL>(p);
            if constexpr (sizeof(T) == 4) { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
            else { dst[0] = widen<T>(v[0]); dst[1] = widen<T>(v[1]); dst[2] = widen<T>(v[2]); dst[3] = widen<T>(v[3]); }
            row_loader<T, N - 4, NTL>::run(p + 4, dst + 4);
        } else if constexpr (N >= 2) {
            typename vec_of<T, 2>::type v = load_vec<typename vec_of<T, 2>::type, NTL>(p);
            if constexpr (sizeof(T) == 4) { dst[0] = v[0]; dst[1] = v[1]; }
            else { dst[0] = widen<T>(v[0]); dst[1] = widen<T>(v[1]); }
            row_loader<T, N - 2, NTL>::run(p + 2, dst + 2);
        } else if constexpr (N == 1) {
            dst[0] = sgv_traits<T>::load(p);
        }
    }
};

template <typename T, int UPX, int UPY, int DOWNX, int DOWNY, int FWP, int FHP, int R0X, int R0Y, int XTRA>
__global__ __launch_bounds__(256) void upfirdn2d_rows_kernel(rows_params p) {
    constexpr int TX = FWP / UPX;                                  // taps per output along x
    constexpr int TY = FHP / UPY;                                  // taps per output along y
    constexpr int NOUT = VEC + XTRA;                               // outputs computed per lane per row
    constexpr int NEED = (R0X + (NOUT - 1) * DOWNX) / UPX + TX;    // input columns per lane
    constexpr int G = UPY;                                         // output rows per loop iteration
    constexpr int ADV = DOWNY * G / UPY;                           // input rows consumed per iteration
    constexpr int WR = (R0Y + (G - 1) * DOWNY) / UPY + TY;         // window rows
    static_assert(FWP % UPX == 0 && FHP % UPY == 0, "padded filter must be a multiple of up");
    static_assert(WR >= ADV, "window smaller than advance");

    const int lane = threadIdx.x & 63;
    // Flipped, zero-padded filter taps (upfirdn2d.cu:118-130): lane t < 16 fetches tap (t/4, t%4), v_readlane
    // broadcasts the 16 values into SGPRs.  Done first, while every lane of the wave is still active -- readlane
    // must only read lanes that executed the load.
    float ff[FHP][FWP];
    {
        static_assert(FHP * FWP <= 64, "one tap per lane");
        const int ta = lane / FWP, tb = lane % FWP;
        float t = 0.f;
        if (lane < FHP * FWP && ta < p.f_h && tb < p.f_w) {
            const int fa = p.flip ? ta : p.f_h - 1 - ta;
            const int fb = p.flip ? tb : p.f_w - 1 - tb;
            t = p.f[fa * p.f_sh + fb * p.f_sw];
        }
#pragma unroll
        for (int a = 0; a < FHP; a++)
#pragma unroll
            for (int b = 0; b < FWP; b++) ff[a][b] = lane_bcast(t, a * FWP + b);
    }

    // readfirstlane: the wave index is uniform, but only provably so to the compiler this way; everything
    // derived from it (strip bounds, row validity) then lives in SGPRs and branches on SCC.
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // wave id -> (plane group, column group, strip); strips fastest so the waves of a workgroup
    // own consecutive strips of one plane group (their halo rows meet in L1/L2).
    const int strip = wave % p.strips;
    const int cg = (wave / p.strips) % p.col_groups;
    const int pg = wave / (p.strips * p.col_groups);
    if (pg >= p.plane_groups) return;

    const int lpr = 1 << p.lpr_log2;
    const int plane = pg * (64 >> p.lpr_log2) + (lane >> p.lpr_log2);
    const int cb = cg * lpr + (lane & (lpr - 1));
    const int ox0 = cb * VEC;
    const int n_main = p.xtra ? p.out_w - 1 : p.out_w;  // columns covered by the VEC-wide blocks
    if (plane >= p.planes || ox0 >= n_main) return;

    const T* xplane = (const T*)p.x + (size_t)plane * p.in_h * p.in_w;
    T* yplane = (T*)p.y + (size_t)plane * p.out_h * p.out_w;

    // First input column of this lane's window; exact because ox0*DOWNX is a multiple of UPX.
    const int inx0 = (ox0 * DOWNX + UPX - 1 - p.pad_x0 - R0X) / UPX;  // numerator divisible by UPX
    const bool cols_inside = (inx0 >= 0) && (inx0 + NEED <= p.in_w);

    const int oy_a = strip * p.strip_h;
    const int oy_b = min(oy_a + p.strip_h, p.out_h);
    if (oy_a >= oy_b) return;
    const int iny0 = (oy_a * DOWNY + UPY - 1 - p.pad_y0 - R0Y) / UPY;  // divisible as well

    float win[WR][NEED];

    auto load_row = [&](int iy, float* dst) {
        if (iy < 0 || iy >= p.in_h) {
#pragma unroll
            for (int i = 0; i < NEED; i++) dst[i] = 0.f;
            return;
        }
        const T* row = xplane + (size_t)iy * p.in_w;
        if (cols_inside) {
            row_loader<T, NEED>::run(row + inx0, dst);
        } else {
#pragma unroll
            for (int i = 0; i < NEED; i++) {
                int c = inx0 + i;
                dst[i] = (c >= 0 && c < p.in_w) ? sgv_traits<T>::load(row + c) : 0.f;
            }
        }
    };

    // Prologue: rows that the first iteration does not load itself.
#pragma unroll
    for (int r = 0; r < WR - ADV; r++) load_row(iny0 + r, win[ADV + r]);

    const bool store_vec = (ox0 + VEC <= n_main);
    const bool store_xtra = XTRA && (ox0 + VEC == p.out_w - 1);

    int iy_next = iny0 + (WR - ADV);
#pragma unroll 4
    for (int oy = oy_a; oy < oy_b; oy += G) {
        // Slide the window down by ADV rows and fetch the new rows.
#pragma unroll
        for (int r = 0; r < WR - ADV; r++)
#pragma unroll
            for (int i = 0; i < NEED; i++) win[r][i] = win[r + ADV][i];
#pragma unroll
        for (int q = 0; q < ADV; q++) load_row(iy_next + q, win[WR - ADV + q]);
        iy_next += ADV;

#pragma unroll
        for (int u = 0; u < G; u++) {
            if (oy + u >= oy_b) break;
            constexpr int dummy = 0; (void)dummy;
            const int ry = (R0Y + u * DOWNY) / UPY;
            const int fy0 = UPY - 1 - ((R0Y + u * DOWNY) % UPY);
            float out[NOUT];
#pragma unroll
            for (int v = 0; v < NOUT; v++) {
                const int cx = (R0X + v * DOWNX) / UPX;
                const int fx0 = UPX - 1 - ((R0X + v * DOWNX) % UPX);
                float acc = 0.f;
#pragma unroll
                for (int ky = 0; ky < TY; ky++)
#pragma unroll
                    for (int kx = 0; kx < TX; kx++)
                        acc = __builtin_fmaf(win[ry + ky][cx + kx], ff[fy0 + ky * UPY][fx0 + kx * UPX], acc);
                out[v] = acc * p.gain;
            }
            T* yrow = yplane + (size_t)(oy + u) * p.out_w + ox0;
            if (store_vec) {
                typename vec_of<T, VEC>::type sv;
                if constexpr (sizeof(T) == 4) { sv[0] = out[0]; sv[1] = out[1]; sv[2] = out[2]; sv[3] = out[3]; }
                else { sv[0] = narrow<T>(out[0]); sv[1] = narrow<T>(out[1]); sv[2] = narrow<T>(out[2]); sv[3] = narrow<T>(out[3]); }
                *(typename vec_of<T, VEC>::type*)yrow = sv;
            } else {
#pragma unroll
                for (int v = 0; v < VEC; v++)
                    if (ox0 + v < n_main) sgv_traits<T>::store(yrow + v, out[v]);
            }
            if constexpr (XTRA) {
                if (store_xtra) sgv_traits<T>::store(yrow + VEC, out[VEC]);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Lane-exchange kernel: the hot calls at >= 129 output columns (one wave spans a whole row block).
//
// Same row walk as the row-walker above, but every input element is loaded from memory exactly once per
// strip: a lane loads only the OWN = VEC*DOWN/UP input columns directly under its VEC output columns (one
// dwordx4 / dwordx2 / 2x dwordx4, 16-B aligned relative to the row start); the L columns it needs to the left and
// the R columns to the right come out of the neighbouring lanes' registers through DPP wave shifts
// (v_mov_b32 wave_shr:1 / wave_shl:1, full-rate VALU, no LDS).  The wave's own outer halo (L + R columns per row)
// is fetched by ONE masked dword load (lanes 0..R-1: right halo, lanes 32..32+L-1: left halo) and injected into
// lane 63 / lane 0 with v_readlane.  Padding and the filter phase are template parameters (the four hot-path
// geometries), so L, R and every window index are compile-time constants.  Input rows are software-pipelined:
// the loads of the next DEPTH row groups are in flight while the current group is filtered, and the outputs,
// which nothing re-reads soon, are written with non-temporal stores.  Measured on MI355X for
// [32,64,257,257]->[32,64,256,256] fp32: 5.4 TB/s vs 3.2 TB/s for the row-walker (a plain float4 copy of the same
// bytes with the same short-strip structure reaches 6.2 TB/s).

struct lanes_params {
    const void* x;
    const float* f;
    void* y;
    int flip;
    float gain;
    int in_w, in_h, out_w, out_h;
    int planes;
    int f_w, f_h;
    int64_t f_sw, f_sh;
    int col_groups;
    int strips;
    int strip_h;
    int lpr_log2;      // SEG only: lanes per image row = 1 << lpr_log2 (<= 32), 64 >> lpr_log2 planes share a wave
    int plane_groups;  // SEG: groups of planes sharing a wave; otherwise == planes
    int nt_store;      // stream the output past the caches (tensors larger than the Infinity Cache)
    int lds_share;     // hand the vertical halo rows from wave to wave through LDS
    // fused epilogue / prologue (EPI != 0), see sgv_fir_epilogue in include/sgv_ops.h
    const float* ep_scale;  // [planes] or NULL
    const float* ep_bias;   // [chans] or NULL
    const void* ep_yref;    // EPI == 2: forward output, same shape as x
    float* ep_sum_g;        // EPI == 2: [planes]
    float* ep_sum_gv;       // EPI == 2: [planes]
    int ep_act;             // 1 linear, 3 lrelu
    float ep_alpha, ep_gain, ep_clamp;
    int chans;
    float* y_amax;          // fp32 tensors: max |y| as a by-product (sgv_amax_sink; positions a lane computes but does not store count too: still an upper bound), or NULL
};

template <typename T, int N> __device__ __forceinline__ void store_vec_nt(T* p, const float* v) {
    typename vec_of<T, N>::type sv;
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (sizeof(T) == 4) sv[i] = v[i]; else sv[i] = narrow<T>(v[i]);
    }
    __builtin_nontemporal_store(sv, (typename vec_of<T, N>::type*)p);
}

template <typename T, int N> __device__ __forceinline__ void store_vec_plain(T* p, const float* v) {
    typename vec_of<T, N>::type sv;
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (sizeof(T) == 4) sv[i] = v[i]; else sv[i] = narrow<T>(v[i]);
    }
    *(typename vec_of<T, N>::type*)p = sv;
}

// EPI = 0: plain upfirdn2d.
// EPI = 1: fused forward epilogue   y = clamp(act(upfirdn2d(x) * scale[n,c] + bias[c]) * gain)  -- the FIR, the
//          demodulation scaling and the bias/activation/clamp of a StyleGAN synthesis layer in ONE pass over the
//          activation instead of three (networks.py:65-74,141-143 + conv2d_resample.py:138-139).
// EPI = 3: fused backward epilogue  dx = d(bias_act)/dx at yref (OUTPUT-shaped) applied to upfirdn2d(x): the gradient of "activation, then FIR"
//          (a discriminator conv0 whose output feeds the FIR of the down-sampling conv1) -- FIR-transposed pass and activation gradient in one;
//          sum_g[n,c] += sum(dx).
// EPI = 4: y = upfirdn2d(x) + yref (OUTPUT-shaped): a gradient summed into the one that already arrived from the tensor's other consumer
//          (the residual discriminator block: the skip branch's FIR gradient + conv0's data gradient), in the store of the streaming pass.
// EPI = 2: fused backward prologue  dx = upfirdn2d(g * scale),  g = d(bias_act)/dx at yref applied to the incoming
//          gradient (bias_act.cu:60-61,133-142 grad=1 form), evaluated while the rows are loaded; the per-plane sums
//          sum(g) and sum(g * preactivation) that give the bias and scale gradients are accumulated on the way.
template <typename T, int UP, int DOWN, int PX0, int PY0, int XTRA, bool SEG, int WPB, int EPI>
__global__ __launch_bounds__(64 * WPB, (DOWN == 2 ? 4 : (EPI >= 2 ? 6 : 8))) void upfirdn2d_lanes_kernel(lanes_params p) {
    constexpr int FWP = 4, FHP = 4, DEPTH = (UP == 2 ? 2 : 1);  // row groups of loads in flight ahead of the math
    constexpr int TX = FWP / UP, TY = FHP / UP;
    constexpr int R0X = ((UP - 1 - PX0) % UP + UP) % UP, R0Y = ((UP - 1 - PY0) % UP + UP) % UP;
    constexpr int NOUT = VEC + XTRA;
    constexpr int NEED = (R0X + (NOUT - 1) * DOWN) / UP + TX;
    constexpr int OWN = VEC * DOWN / UP;                    // input columns owned (loaded) by a lane
    constexpr int L = -((UP - 1 - PX0 - R0X) / UP);           // columns needed left of the owned block
    constexpr int R = NEED - OWN - L;                        // ... and right of it
    constexpr int G = UP, ADV = DOWN * G / UP;
    constexpr int WR = (R0Y + (G - 1) * DOWN) / UP + TY;
    constexpr int BASEY = (UP - 1 - PY0 - R0Y) / UP;          // input row of window row 0 for output row 0
    static_assert(L >= 0 && R >= 0 && L <= OWN && R <= OWN && L < 32 && R < 32, "geometry not supported by the lane-exchange kernel");
    static_assert(WR >= ADV, "window smaller than advance");

    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * WPB + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = wave % p.strips;
    // SEG = false: the wave spans one block of 64*VEC output columns of one plane (column group cg).
    // SEG = true : rows are at most 32 lanes wide; 64 >> lpr_log2 planes ride in one wave, lane groups exchange
    //              only inside their own plane and the last lane of a group fetches its right halo itself.
    const int cg = SEG ? 0 : (wave / p.strips) % p.col_groups;
    const int pgroup = SEG ? wave / p.strips : wave / (p.strips * p.col_groups);
    const int lpr = SEG ? (1 << p.lpr_log2) : 64;
    const int sub = lane & (lpr - 1);
    const int plane = SEG ? pgroup * (64 >> p.lpr_log2) + (lane >> p.lpr_log2) : pgroup;
    // Waves past the end of the grid still take part in the workgroup barrier below, then leave.
    const bool wave_active = pgroup < (SEG ? p.plane_groups : p.planes);
    const bool plane_ok = wave_active && plane < p.planes;
    const bool seg_first = SEG && sub == 0, seg_last = SEG && sub == lpr - 1;

    float ff[FHP][FWP];
    {   // lane t < 16 fetches tap (t/4, t%4); v_readlane broadcasts the 16 values into SGPRs
        const int ta = (lane >> 2) & 3, tb = lane & 3;
        float t = 0.f;
        if (lane < 16 && ta < p.f_h && tb < p.f_w) {
            const int fa = p.flip ? ta : p.f_h - 1 - ta;
            const int fb = p.flip ? tb : p.f_w - 1 - tb;
            t = p.f[fa * p.f_sh + fb * p.f_sw];
        }
#pragma unroll
        for (int a = 0; a < FHP; a++)
#pragma unroll
            for (int b = 0; b < FWP; b++) ff[a][b] = lane_bcast(t, a * 4 + b);
    }

    const T* xplane = (const T*)p.x + (size_t)(plane_ok ? plane : 0) * p.in_h * p.in_w;
    T* yplane = (T*)p.y + (size_t)(plane_ok ? plane : 0) * p.out_h * p.out_w;
    const int cb = SEG ? sub : cg * 64 + lane;
    const int ox0 = cb * VEC;
    float ep_sc = 1.f, ep_bi = 0.f;
    if constexpr (EPI != 0) {
        if (plane_ok && p.ep_scale) ep_sc = p.ep_scale[plane];
        if (plane_ok && p.ep_bias) ep_bi = p.ep_bias[plane % p.chans];
    }
    const T* yrefplane = (EPI == 2) ? (const T*)p.ep_yref + (size_t)(plane_ok ? plane : 0) * p.in_h * p.in_w
                       : (EPI == 3 || EPI == 4) ? (const T*)p.ep_yref + (size_t)(plane_ok ? plane : 0) * p.out_h * p.out_w : nullptr;
    float sum_g = 0.f, sum_gv = 0.f;
    const int own0 = cb * OWN;                               // first owned input column
    const bool own_full = plane_ok && own0 + OWN <= p.in_w;
    const bool own_none = !plane_ok || own0 >= p.in_w;
    // outer halo of the wave (SEG = false): lanes [0,R) fetch the columns right of lane 63's block, lanes [32,32+L)
    // those left of lane 0's.  With SEG the left halo is always zero padding (single column group).
    int halo_col = -1;
    if (!SEG && lane < R) halo_col = (cg * 64 + 64) * OWN + lane;
    if (!SEG && lane >= 32 && lane < 32 + L) halo_col = cg * 64 * OWN - L + (lane - 32);
    const bool halo_ok = halo_col >= 0 && halo_col < p.in_w;
    const int seg_halo0 = own0 + OWN;  // first right-halo column of a group-last lane

    const int n_main = XTRA ? p.out_w - 1 : p.out_w;
    const int oy_a = strip * p.strip_h;
    const int oy_b = min(oy_a + p.strip_h, p.out_h);       // strips = ceil(out_h / strip_h): never empty
    const int iny0 = oy_a * DOWN / UP + BASEY;               // strip_h is a multiple of UP
    const int iy_last = ((oy_b - 1) * DOWN + UP - 1 - PY0) / UP + TY;  // (generous) last input row any output of the strip touches

    // Vertical-halo hand-off through LDS.  The last WR-ADV input rows of a strip are exactly the first rows the next
    // strip loads in its prologue; the next strip belongs to the next wave of this workgroup, which fetches them at
    // t = 0 while this wave needs them only at the very end of its walk -- too far apart for L1/L2 to help (measured:
    // 1.24x the algorithmic read bytes on the headline call).  So every wave parks its prologue rows (raw, as loaded)
    // in LDS, one barrier, and the wave above reads them from there instead of from memory.
    constexpr int SHARE = WR - ADV;                          // rows shared with the strip above
    constexpr int HR = SEG ? (R > 0 ? R : 1) : 1;
    constexpr int RAW = OWN + 1 + HR;
    __shared__ float halo_lds[SHARE > 0 ? WPB * SHARE * RAW * 64 : 1];
    const int wslot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the wave below is this strip's successor iff it is in the same plane group / column group (strips run fastest)
    // (and only if this strip is tall enough that its own prologue rows are not already the shared ones: those are read
    // before the barrier)
    const bool lower_in_wg = SHARE > 0 && p.lds_share && wave_active && wslot < WPB - 1 && strip + 1 < p.strips &&
                             ((oy_b - oy_a + G - 1) / G) * ADV >= SHARE;
    const int shared_row0 = oy_b * DOWN / UP + BASEY;        // == iny0 of the next strip when this strip is full-height

    struct raw_row { float m[OWN]; float h; float hr[HR]; };

    // rows this strip accounts for in the plane sums (every input row exactly once over the strips of a plane)
    const int own_lo = iny0;
    const int own_hi = (strip + 1 < p.strips) ? shared_row0 : 0x7fffffff;
    // bias -> activation -> gain -> clamp, the operation order of bias_act.cu:51-142 (grad 0)
    auto epi_fwd = [&](float u) {
        float t = u * ep_sc;
        t = t + ep_bi;
        if (p.ep_act == 3) t = (t > 0.f) ? t : t * p.ep_alpha;
        t *= p.ep_gain;
        if (p.ep_clamp >= 0.f) t = (t > -p.ep_clamp & t < p.ep_clamp) ? t : (t >= 0.f) ? p.ep_clamp : -p.ep_clamp;
        return t;
    };
    // grad-1 form (bias_act.cu:60-61,133-142): g = dy * act'(pre) * gain, zero where the forward output was clamped;
    // `pre` returns the pre-activation value recovered from the stored output (lrelu is invertible).
    // (round 6: y / gain and y / alpha as products with reciprocals taken once per thread -- two IEEE divisions per ELEMENT were a third of the prologue's
    // instructions; the sign test is unchanged, the recovered pre-activation moves by <= 1 ulp and only enters the plane sum of g * pre)
    const float ep_inv_gain = (p.ep_gain != 0.f) ? 1.f / p.ep_gain : 0.f, ep_inv_alpha = (p.ep_alpha != 0.f) ? 1.f / p.ep_alpha : 0.f;
    auto epi_grad = [&](float dy, float yref, float& pre) {
        const float yy = yref * ep_inv_gain;
        float g = dy;
        pre = yy;
        if (p.ep_act == 3) { g = (yy > 0.f) ? dy : dy * p.ep_alpha; pre = (yy > 0.f) ? yy : yy * ep_inv_alpha; }
        g *= p.ep_gain;
        if (p.ep_clamp >= 0.f) g = (yref > -p.ep_clamp & yref < p.ep_clamp) ? g : 0.f;
        return g;
    };

    auto lds_put = [&](int r, const raw_row& t) {
        if constexpr (SHARE > 0) {
            float* q = halo_lds + ((wslot * SHARE + r) * RAW) * 64 + lane;
#pragma unroll
            for (int i = 0; i < OWN; i++) q[i * 64] = t.m[i];
            q[OWN * 64] = t.h;
#pragma unroll
            for (int i = 0; i < HR; i++) q[(OWN + 1 + i) * 64] = t.hr[i];
        }
    };
    auto lds_get = [&](int r, raw_row& t) {
        if constexpr (SHARE > 0) {
            const float* q = halo_lds + (((wslot + 1) * SHARE + r) * RAW) * 64 + lane;
#pragma unroll
            for (int i = 0; i < OWN; i++) t.m[i] = q[i * 64];
            t.h = q[OWN * 64];
#pragma unroll
            for (int i = 0; i < HR; i++) t.hr[i] = q[(OWN + 1 + i) * 64];
        }
    };

    auto issue = [&](int iy, raw_row& r) {
#pragma unroll
        for (int i = 0; i < OWN; i++) r.m[i] = 0.f;
        r.h = 0.f;
#pragma unroll
        for (int i = 0; i < HR; i++) r.hr[i] = 0.f;
        if (iy < 0 || iy >= p.in_h || iy > iy_last) return;   // wave-uniform
        if (lower_in_wg && iy >= shared_row0 && iy < shared_row0 + SHARE) { lds_get(iy - shared_row0, r); return; }  // wave-uniform
        const T* row = xplane + (size_t)iy * p.in_w;
        if (own_full) {
            row_loader<T, OWN>::run(row + own0, r.m);
        } else if (!own_none) {
#pragma unroll
            for (int i = 0; i < OWN; i++)
                if (own0 + i < p.in_w) r.m[i] = sgv_traits<T>::load(row + own0 + i);
        }
        if constexpr (SEG) {
            if (seg_last && plane_ok) {
#pragma unroll
                for (int i = 0; i < R; i++)
                    if (seg_halo0 + i < p.in_w) r.hr[i] = sgv_traits<T>::load(row + seg_halo0 + i);
            }
        } else {
            if (halo_ok) r.h = sgv_traits<T>::load(row + halo_col);
        }
        if constexpr (EPI == 2) {
            // The rows just loaded hold the incoming gradient dy; turn them into g * scale with the forward output at
            // the same positions.  Rows in [own_lo, own_hi) belong to this strip for the purpose of the plane sums.
            const T* yrow = yrefplane + (size_t)iy * p.in_w;
            const bool counted = iy >= own_lo && iy < own_hi;
            float yr[OWN];
#pragma unroll
            for (int i = 0; i < OWN; i++) yr[i] = 0.f;
            if (own_full) {
                row_loader<T, OWN>::run(yrow + own0, yr);
            } else if (!own_none) {
#pragma unroll
                for (int i = 0; i < OWN; i++)
                    if (own0 + i < p.in_w) yr[i] = sgv_traits<T>::load(yrow + own0 + i);
            }
#pragma unroll
            for (int i = 0; i < OWN; i++) {
                float pre;
                const float g = epi_grad(r.m[i], yr[i], pre);
                if (counted) { sum_g += g; sum_gv = __builtin_fmaf(g, pre, sum_gv); }
                r.m[i] = g * ep_sc;
            }
            float pre;
            if constexpr (SEG) {
                if (seg_last && plane_ok) {
#pragma unroll
                    for (int i = 0; i < R; i++)
                        if (seg_halo0 + i < p.in_w) r.hr[i] = epi_grad(r.hr[i], sgv_traits<T>::load(yrow + seg_halo0 + i), pre) * ep_sc;
                }
            } else {
                // the outer-halo columns belong to this plane too (non-SEG: one plane per wave), so ep_sc is the right scale
                if (halo_ok) r.h = epi_grad(r.h, sgv_traits<T>::load(yrow + halo_col), pre) * lane_bcast(ep_sc, 0);
            }
        }
    };

    auto expand = [&](const raw_row& r, float* dst) {   // dst[NEED]: columns own0-L .. own0-L+NEED-1
#pragma unroll
        for (int i = 0; i < L; i++) {
            if constexpr (SEG) { float v = dpp_wave_shr1(r.m[OWN - L + i], 0.f); dst[i] = seg_first ? 0.f : v; }
            else dst[i] = dpp_wave_shr1(r.m[OWN - L + i], lane_bcast(r.h, 32 + i));
        }
#pragma unroll
        for (int i = 0; i < OWN; i++) dst[L + i] = r.m[i];
#pragma unroll
        for (int i = 0; i < R; i++) {
            if constexpr (SEG) { float v = dpp_wave_shl1(r.m[i], 0.f); dst[L + OWN + i] = seg_last ? r.hr[i] : v; }
            else dst[L + OWN + i] = dpp_wave_shl1(r.m[i], lane_bcast(r.h, i));
        }
    };

    float win[WR][NEED];
    if (wave_active) {
        raw_row t;
#pragma unroll
        for (int r = 0; r < WR - ADV; r++) { issue(iny0 + r, t); if (p.lds_share) lds_put(r, t); expand(t, win[ADV + r]); }
    }
    if constexpr (SHARE > 0) { if (p.lds_share) __syncthreads(); }
    if (!wave_active) return;
    raw_row ring[DEPTH][ADV];
    int iy_next = iny0 + (WR - ADV);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
        for (int q = 0; q < ADV; q++) issue(iy_next + q, ring[d][q]);
        iy_next += ADV;
    }

    const bool store_vec = plane_ok && (ox0 + VEC <= n_main);
    const bool store_any = plane_ok && ox0 < n_main;
    const bool store_xtra = XTRA && plane_ok && (ox0 + VEC == p.out_w - 1);
    unsigned amx = 0;

    for (int oy = oy_a; oy < oy_b; oy += G * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int oyd = oy + d * G;
            if (oyd >= oy_b) break;
#pragma unroll
            for (int r = 0; r < WR - ADV; r++)
#pragma unroll
                for (int i = 0; i < NEED; i++) win[r][i] = win[r + ADV][i];
#pragma unroll
            for (int q = 0; q < ADV; q++) expand(ring[d][q], win[WR - ADV + q]);
            // refill this ring slot with the rows of the group DEPTH iterations ahead
#pragma unroll
            for (int q = 0; q < ADV; q++) issue(iy_next + q, ring[d][q]);
            iy_next += ADV;

#pragma unroll
            for (int u = 0; u < G; u++) {
                if (oyd + u >= oy_b) break;
                const int ry = (R0Y + u * DOWN) / UP;
                const int fy0 = UP - 1 - ((R0Y + u * DOWN) % UP);
                float out[NOUT];
                float yo[NOUT];
                if constexpr (EPI == 3 || EPI == 4) {   // the forward output / the other summand at this row's positions (same predicates as the stores below)
#pragma unroll
                    for (int v = 0; v < NOUT; v++) yo[v] = 0.f;
                    const T* yr = yrefplane + (size_t)(oyd + u) * p.out_w + ox0;
                    if (store_vec) row_loader<T, VEC>::run(yr, yo);
                    else if (store_any) {
#pragma unroll
                        for (int v = 0; v < VEC; v++)
                            if (ox0 + v < n_main) yo[v] = sgv_traits<T>::load(yr + v);
                    }
                    if constexpr (XTRA) { if (store_xtra) yo[VEC] = sgv_traits<T>::load(yr + VEC); }
                }
#pragma unroll
                for (int v = 0; v < NOUT; v++) {
                    const int cx = (R0X + v * DOWN) / UP;
                    const int fx0 = UP - 1 - ((R0X + v * DOWN) % UP);
                    float acc = 0.f;
#pragma unroll
                    for (int ky = 0; ky < TY; ky++)
#pragma unroll
                        for (int kx = 0; kx < TX; kx++)
                            acc = __builtin_fmaf(win[ry + ky][cx + kx], ff[fy0 + ky * UP][fx0 + kx * UP], acc);
                    out[v] = acc * p.gain;
                    if constexpr (EPI == 1) out[v] = epi_fwd(out[v]);
                    if constexpr (EPI == 4) out[v] += yo[v];
                    if constexpr (EPI == 3) {
                        float pre;
                        out[v] = epi_grad(out[v], yo[v], pre);
                        const bool stored = v < VEC ? (store_vec || (store_any && ox0 + v < n_main)) : store_xtra;
                        if (stored) sum_g += out[v];
                    }
                    if constexpr (sizeof(T) == 4 && EPI != 1) amx = sgv_amax_fold(amx, out[v]);   // (EPI 1: the tile / asm kernels serve the forward epilogue; one more live register spills here)
                }
                T* yrow = yplane + (size_t)(oyd + u) * p.out_w + ox0;
                if (store_vec) {
                    if (p.nt_store) store_vec_nt<T, VEC>(yrow, out);
                    else store_vec_plain<T, VEC>(yrow, out);
                } else if (store_any) {
#pragma unroll
                    for (int v = 0; v < VEC; v++)
                        if (ox0 + v < n_main) sgv_traits<T>::store(yrow + v, out[v]);
                }
                if constexpr (XTRA) {
                    if (store_xtra) sgv_traits<T>::store(yrow + VEC, out[VEC]);
                }
            }
        }
    }
    if constexpr (EPI == 2 || EPI == 3) {
        // reduce over the lanes that share a plane (the whole wave, or one lane group with SEG), then one atomic per plane
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            if (off < lpr) { sum_g += __shfl_xor(sum_g, off, 64); sum_gv += __shfl_xor(sum_gv, off, 64); }
        }
        if (sub == 0 && plane_ok) {
            atomicAdd(p.ep_sum_g + plane, sum_g);
            if (p.ep_sum_gv) atomicAdd(p.ep_sum_gv + plane, sum_gv);
        }
    }
    if constexpr (sizeof(T) == 4 && EPI != 1) { if (p.y_amax) sgv_amax_commit(amx, p.y_amax); }
}

// ---------------------------------------------------------------------------------------------
// Tile kernel: the FIR passes (up = down = 1, filter <= 4x4, pad 0..3) of any width -- the pass after an up-sampling convolution (2r+1 -> 2r),
// the pass in front of a strided one (r -> r+1) and their gradients; plain (EPI = 0), with the synthesis layer's forward epilogue (EPI = 1,
// sgv_upfirdn2d_fused mode 1) or with the backward epilogue of "activation, then FIR" (EPI = 3, mode 3).
//
// Round 3 (tools/ufd_lab.hip V6, profiles/r03_ufd_lab_tile.log): a workgroup of 4 waves owns 16 output rows x 64 lanes x 4 (+1) output columns.
// LOAD phase: the 19 input rows of the tile are dealt round-robin to the waves and EVERY load of the workgroup leaves at t = 0 -- per row one
// 16-byte load per lane (the window clamped into the row, shifted back with selects) plus the 3 (+1) right-halo columns by the first lanes of a
// row -- exactly the request pattern of the float4 copy that reaches 6.2-6.4 TB/s on this part, instead of a strip walk that interleaves
// loads, arithmetic and stores for the lifetime of a wave (lanes kernel 5.0-5.3, asm-load strip kernel 5.4-5.6 on the headline call; this
// kernel 5.8-6.0).  The rows are parked, zero padding included, in LDS; one barrier.  COMPUTE phase: wave w produces output rows 4w .. 4w+3
// from LDS rows 4w .. 4w+6 (two aligned ds_read_b128 per lane and row: its own four columns and the next four), one fmaf chain per output in the
// reference's tap order (rows, then columns) -- bit-identical to the oracle like every other path -- and streams the rows out.  No inline
// asm, no counted waits: nothing here depends on what the compiler does with registers an asm load has not yet written.
// Narrow planes (<= 128 columns): 64 >> lpr_log2 planes sit side by side in the 64 lanes, each with its own halo words in the LDS row.
struct tile_params {
    const void* x;
    const float* f;
    void* y;
    int flip;
    float gain;
    int in_w, in_h, out_w, out_h;
    int planes;
    int f_w, f_h;
    int64_t f_sw, f_sh;
    int pad_x, pad_y;
    int lpr_log2;      // lanes per plane row = 1 << lpr_log2 (64: one plane per wave row, col_groups column groups of 256 outputs)
    int col_groups;
    int row_tiles;     // ceil(out_h / 16)
    int xcd_blocks;    // workgroups [0, xcd_blocks) are dealt so that the row tiles of one (plane group, column group) stay on ONE XCD; a multiple of 8 * row_tiles
    int lds_amax_word; // float index of the workgroup's running-maximum word behind the tile image
    int nt_store;
    const float* ep_scale;  // EPI 1: [planes] or NULL
    const float* ep_bias;   // EPI 1: [chans] or NULL
    const void* ep_yref;    // EPI 3: the forward output, OUTPUT-shaped; EPI 2: the forward output of the layer whose gradient arrives, INPUT-shaped
    float* ep_sum_g;        // EPI 3: [planes], += sum of the result over the plane; EPI 2: += sum of g (the activation gradient, before the scale)
    float* ep_sum_gv;       // EPI 2: [planes], += sum of g * pre-activation
    int ep_act;
    float ep_alpha, ep_gain, ep_clamp;
    int chans;
    float* y_amax;          // fp32 tensors: max |y| as a by-product (sgv_amax_sink), or NULL
};

constexpr int TILE_ROWS = 16, TILE_IN_ROWS = TILE_ROWS + 3;
inline int tile_lds_floats(int lpr_log2, int cpl = 4) { return TILE_IN_ROWS * (64 >> lpr_log2) * ((cpl << lpr_log2) + 8); }

// WIDE: one plane per wave row (lpr_log2 == 6): the LDS pitches are compile-time constants (the instantiation of the >= 129-column calls, which
// carry most of the bytes; the run-time-pitch form spent ~30 % more instructions on addresses and measured 6-8 % below tools/ufd_lab.hip V6).
// NT: stream the output past the caches (tensors larger than the Infinity Cache).  A template parameter, not a branch: with both store forms in the
//     kernel -- even as two copies of the row loop behind one wave-uniform branch -- the headline call ran 5 % slower (tools/ufd_lab.hip
//     SGV_TILE_EXP=8, profiles/r03_ufd_tile_bisect.log).
// F44: the filter is a dense 4 x 4 fp32 array (what setup_filter produces): its 16 taps are ONE s_load_dwordx16 issued with the kernel arguments;
//     sixteen separately addressed scalar loads (any size <= 4 x 4, any strides: F44 = false) cost 3-4 % on the same call.
// CPL: output columns per lane.  4 for fp32 (one 16-byte request per lane and row).  8 for the 16-bit formats (round 3, second part): with four columns a
//     16-bit lane asks for 8 bytes per row and the pass ran at the fp32 ELEMENT rate, i.e. half the bytes per second (bf16 [96,64,257,257]: 855 us
//     against 913 us in fp32, profiles/r03_bench_step_lowp_bf16_kernel_stats.csv); with eight it issues the same 16-byte requests as the fp32 form.
//     The LDS tile holds fp32 either way (19 rows x (64 x CPL + 8) floats).
// Round 6 (tools/ufd_lab6.hip, profiles/r06_c1_ufd_lab6.log):
//  * XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin (blockIdx % 8) and each XCD has its own L2: with the row tiles of a plane on consecutive
//    block indices the 3 halo rows two neighbouring tiles share were fetched through two different L2s, i.e. twice from the fabric (19 / 16 of the read
//    bytes).  Now the 8 XCDs walk 8 different (plane group, column group) units, each through all its row tiles: +1-4 % (5.83 -> 5.89 TB/s at 32 frames,
//    5.93 -> 6.17 at 96 on the lab form of this kernel).
//  * The 12 input rows of a tile that no other tile reads are fetched with non-temporal requests; the shared ones keep the default policy (they are the lines
//    the L2 should hold).
//  * Measured and NOT kept (profiles/r06_c4_ufd_lab6_steady_state_ablation.log): (i) naturally aligned 16-byte stores for outputs of pitch 4 k + 1 (256 -> 257) built
//    with DPP wave shifts plus dword stores by lanes 0 / 63: 7 % SLOWER than the misaligned 16-byte store + one dword (194 vs 180 us at 32 frames), although a
//    float4 copy with a misaligned destination loses 7 %; the same through an LDS-staged linear span (tools/ufd_lab6.hip V8 stage1): +2-3 %, not worth a second
//    barrier pair; (ii) a 64-register bound (8 waves per SIMD): 0 ... -2.7 %.  The window is still read in two steps (40 live window registers instead of 56).
template <typename T, int XTRA, int EPI, bool WIDE, bool NT, bool F44, int CPL = 4>
#ifndef SGV_TILE_LAB_OFF
#define SGV_TILE_LAB_OFF 0      // lab builds (tools/ufd_lab6.hip): 2 = no non-temporal loads, 16 = scalar multiply-add chains, 32 = no running maximum
#endif
__global__ __launch_bounds__(256) void upfirdn2d_tile_kernel(tile_params p) {
    constexpr int NH = 3 + XTRA, NOUT = CPL + XTRA;
    static_assert(CPL == 4 || CPL == 8, "4 or 8 output columns per lane");
    extern __shared__ __attribute__((aligned(16))) float tile_lds[];
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rt, unit;
    if ((int)blockIdx.x < p.xcd_blocks) { const int j = blockIdx.x >> 3; rt = j % p.row_tiles; unit = (j / p.row_tiles) * 8 + (blockIdx.x & 7); }
    else { rt = blockIdx.x % p.row_tiles; unit = blockIdx.x / p.row_tiles; }
    const int cg = unit % p.col_groups;
    const int pg = unit / p.col_groups;
    const int lpr = WIDE ? 64 : 1 << p.lpr_log2;
    const int sub = WIDE ? lane : lane & (lpr - 1), slot = WIDE ? 0 : lane >> p.lpr_log2;
    const int plane = WIDE ? pg : pg * (64 >> p.lpr_log2) + slot;
    const bool plane_ok = plane < p.planes;
    const int seg_pitch = CPL * lpr + 8;                      // floats of one plane's row in LDS: CPL per lane + the halo words (16-byte aligned)
    const int row_pitch = WIDE ? 64 * CPL + 8 : (64 >> p.lpr_log2) * seg_pitch;
    float ff[4][4];
    if constexpr (F44) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) ff[a][b] = p.flip ? p.f[a * 4 + b] : p.f[(3 - a) * 4 + (3 - b)];
    }
    const T* xp = (const T*)p.x + (size_t)(plane_ok ? plane : 0) * p.in_h * p.in_w;
    T* yp = (T*)p.y + (size_t)(plane_ok ? plane : 0) * p.out_h * p.out_w;
    float ep_sc = 1.f, ep_bi = 0.f;     // EPI 1: issued in front of the row loads (round 6; behind the barrier their latency sat in front of the first multiply-add)
    if constexpr (EPI == 1) {
        if (plane_ok && p.ep_scale) ep_sc = p.ep_scale[plane];
        if (plane_ok && p.ep_bias) ep_bi = p.ep_bias[plane % p.chans];
    }
    const int n_main = XTRA ? p.out_w - 1 : p.out_w;           // a multiple of CPL
    const int ox = (cg * 64 + sub) * CPL;
    const int ix0 = ox - p.pad_x;
    const int base = min(max(ix0, 0), p.in_w - CPL);            // the CPL-column load window, clamped into the row
    const int sh = base - ix0;                                  // how far it moved: pad_x for the lane at ox = 0, -over_u for the lane that overhangs the row's end
    const int over_u = (((-p.pad_x - p.in_w) % CPL) + CPL) % CPL;   // ix0 + CPL - in_w of that lane (ox is a multiple of CPL): wave-uniform, 0 = no such lane
    const bool cols_dead = !plane_ok || ix0 >= p.in_w || ix0 + CPL - 1 < 0;
    const int ixh = (cg * 64 + lpr) * CPL - p.pad_x + sub;      // lanes sub < NH of a plane row: the columns right of its last lane's block
    const bool halo_ok = plane_ok && sub < NH && ixh >= 0 && ixh < p.in_w;
    const int ixh_c = min(max(ixh, 0), p.in_w - 1);
    const int oy0 = rt * TILE_ROWS;
    const int iy0 = oy0 - p.pad_y;
    const bool st_vec = plane_ok && ox < n_main;
    const bool st_xtra = XTRA && plane_ok && ox + CPL == p.out_w - 1;

    // ---- load phase: rows wave, wave + 4, ... of the 19; everything is issued before anything is used ----
    constexpr int RPW = (TILE_IN_ROWS + 3) / 4;
    float m[RPW][CPL], h[RPW];
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wave + 4 * k;
        const T* row = xp + (size_t)min(max(iy0 + r, 0), p.in_h - 1) * p.in_w;
#pragma unroll
        for (int i = 0; i < CPL; i++) m[k][i] = 0.f;
        h[k] = 0.f;
        if (r < TILE_IN_ROWS) {
            if (k >= 1 && k <= 3 && !(SGV_TILE_LAB_OFF & 2)) row_loader<T, CPL, true>::run(row + base, m[k]);     // tile rows 4 .. 15: read by this tile only
            else row_loader<T, CPL>::run(row + base, m[k]);
            if (sub < NH) h[k] = sgv_traits<T>::load(row + ixh_c);
        }
    }
    // EPI 2 (round 6; the lane-exchange kernel served it at 4.5 TB/s): the rows hold the incoming gradient dy; the forward output at the same positions turns
    // them into g = d(bias_act)/dx . dy before they are parked (times the plane's scale), and every input element is counted ONCE in the plane sums: by the
    // lane that owns its column (not the neighbour whose clamped window also covers it), in the tile whose first 16 rows hold it (the last tile: all 19).
    float yv[EPI == 2 ? RPW : 1][CPL], hy[EPI == 2 ? RPW : 1];
    if constexpr (EPI == 2) {
        const T* yrp = (const T*)p.ep_yref + (size_t)(plane_ok ? plane : 0) * p.in_h * p.in_w;
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            const int r = wave + 4 * k;
            const T* yrow = yrp + (size_t)min(max(iy0 + r, 0), p.in_h - 1) * p.in_w;
#pragma unroll
            for (int i = 0; i < CPL; i++) yv[k][i] = 0.f;
            hy[k] = 0.f;
            if (r < TILE_IN_ROWS) {
                row_loader<T, CPL, true>::run(yrow + base, yv[k]);
                if (sub < NH) hy[k] = sgv_traits<T>::load(yrow + ixh_c);
            }
        }
    }
    float yo[4][NOUT];
    if constexpr (EPI == 3) {   // the forward output at this wave's four output rows (same predicates as the stores)
        const T* yrp = (const T*)p.ep_yref + (size_t)(plane_ok ? plane : 0) * p.out_h * p.out_w;
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int v = 0; v < NOUT; v++) yo[k][v] = 0.f;
            const int oy = min(oy0 + 4 * wave + k, p.out_h - 1);
            const T* yr = yrp + (size_t)oy * p.out_w + ox;
            if (st_vec) row_loader<T, CPL>::run(yr, yo[k]);
            if constexpr (XTRA) { if (st_xtra) yo[k][CPL] = sgv_traits<T>::load(yr + CPL); }
        }
    }
    // Any other filter: the taps AFTER the row loads have been issued, through the scalar cache (every index is wave-uniform).  (The lanes kernel's
    // "16 lanes load, v_readlane broadcasts" makes hipcc wait for that vector load -- vmcnt(0) -- BEFORE the row loads are issued.)
    if constexpr (!F44) {
        const int fsh = (int)p.f_sh, fsw = (int)p.f_sw;          // a filter is at most 4 x 4: 32-bit index arithmetic on the scalar unit
        const int a0 = p.flip ? 0 : (p.f_h - 1) * fsh, da = p.flip ? fsh : -fsh;
        const int b0 = p.flip ? 0 : (p.f_w - 1) * fsw, db = p.flip ? fsw : -fsw;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const bool live = a < p.f_h && b < p.f_w;
                const float v = p.f[live ? a0 + a * da + b0 + b * db : 0];
                ff[a][b] = live ? v : 0.f;
            }
    }
    float sum_g2 = 0.f, sum_gv2 = 0.f, ep_sc2 = 1.f;
    if constexpr (EPI == 2) { if (plane_ok && p.ep_scale) ep_sc2 = p.ep_scale[plane]; }
    // y / gain and y / alpha as products with reciprocals taken once per thread (the sign test is unchanged; the recovered pre-activation moves by <= 1 ulp)
    const float inv_gain = (p.ep_gain != 0.f) ? 1.f / p.ep_gain : 0.f, inv_alpha = (p.ep_alpha != 0.f) ? 1.f / p.ep_alpha : 0.f;
    if (threadIdx.x == 0) tile_lds[p.lds_amax_word] = 0.f;      // the workgroup's running maximum (sgv_amax_commit_wg), zeroed in front of the barrier
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wave + 4 * k;
        if (r >= TILE_IN_ROWS) continue;
        const int iy = iy0 + r;
        const bool row_ok = iy >= 0 && iy < p.in_h;
        // Undo the clamp: column ix0 + i = loaded[i - sh] where that exists, else 0 (padding).  Only the first lane of a plane row (sh = pad_x) and the one
        // lane whose window overhangs the row's end (sh = -over_u) moved, and both amounts are wave-uniform: a uniform branch picks the shift, every
        // element costs one select per side (the per-lane chain over |sh| = 1..3 cost six -- and covered overhangs up to 3 only, short of what
        // CPL = 8 can meet on rows whose column blocks do not fill the lanes).
        if constexpr (EPI == 2) {
            const bool counted = row_ok && plane_ok && (r < TILE_ROWS || rt == p.row_tiles - 1);
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                const float dy = m[k][i], yref = yv[k][i];
                const float yy = yref * inv_gain;
                float g = dy, pre = yy;
                if (p.ep_act == 3) { g = (yy > 0.f) ? dy : dy * p.ep_alpha; pre = (yy > 0.f) ? yy : yy * inv_alpha; }
                g *= p.ep_gain;
                if (p.ep_clamp >= 0.f) g = (yref > -p.ep_clamp & yref < p.ep_clamp) ? g : 0.f;
                if (counted && i + sh >= 0 && i + sh < CPL) { sum_g2 += g; sum_gv2 = __builtin_fmaf(g, pre, sum_gv2); }
                m[k][i] = g * ep_sc2;
            }
            {
                const float dy = h[k], yref = hy[k];
                const float yy = yref * inv_gain;
                float g = dy, pre = yy;
                if (p.ep_act == 3) { g = (yy > 0.f) ? dy : dy * p.ep_alpha; pre = (yy > 0.f) ? yy : yy * inv_alpha; }
                g *= p.ep_gain;
                if (p.ep_clamp >= 0.f) g = (yref > -p.ep_clamp & yref < p.ep_clamp) ? g : 0.f;
                if (counted && halo_ok && cg == p.col_groups - 1) { sum_g2 += g; sum_gv2 = __builtin_fmaf(g, pre, sum_gv2); }
                h[k] = g * ep_sc2;
            }
        }
        float o[CPL];
#pragma unroll
        for (int i = 0; i < CPL; i++) o[i] = m[k][i];
#if !(defined(SGV_TILE_ABL) && (SGV_TILE_ABL & 2))  // lab build: without the selects
#pragma unroll
        for (int d = 1; d <= 3; d++)
            if (p.pad_x == d) {
#pragma unroll
                for (int i = 0; i < CPL; i++) o[i] = (sh > 0) ? (i - d >= 0 ? m[k][i - d] : 0.f) : o[i];
            }
#pragma unroll
        for (int d = 1; d < CPL; d++)
            if (over_u == d) {
#pragma unroll
                for (int i = 0; i < CPL; i++) o[i] = (sh < 0) ? (i + d < CPL ? m[k][i + d] : 0.f) : o[i];
            }
#endif
#pragma unroll
        for (int i = 0; i < CPL; i++) o[i] = (row_ok && !cols_dead) ? o[i] : 0.f;
        float* lrow = tile_lds + r * row_pitch + slot * seg_pitch;
#pragma unroll
        for (int q = 0; q < CPL / 4; q++) *(f4v*)(lrow + CPL * sub + 4 * q) = f4v{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
        if (sub < 8) lrow[CPL * lpr + sub] = (row_ok && halo_ok) ? h[k] : 0.f;   // halo words; the rest of the 8 are zero
    }
    __syncthreads();
    if constexpr (EPI == 2) {   // the plane sums of this wave's rows: reduce over the lanes of a plane row, one atomic pair per plane and wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
            if (off < lpr) { sum_g2 += __shfl_xor(sum_g2, off, 64); sum_gv2 += __shfl_xor(sum_gv2, off, 64); }
        if (sub == 0 && plane_ok) { atomicAdd(p.ep_sum_g + plane, sum_g2); atomicAdd(p.ep_sum_gv + plane, sum_gv2); }
    }

    // ---- compute phase: output rows 4 wave .. 4 wave + 3 from LDS rows 4 wave .. 4 wave + 6 ----
    // (the window is read in two steps -- rows 0 .. 4 in front of output rows 0 and 1, rows 5 and 6 behind them, where rows 0 and 1 are dead -- so that 40
    // window registers are live instead of 56)
    float win[7][CPL + 4];
    auto read_win = [&](int r) {
        const float* lrow = tile_lds + (4 * wave + r) * row_pitch + slot * seg_pitch + CPL * sub;
#pragma unroll
        for (int q = 0; q < CPL / 4 + 1; q++) {     // its own CPL columns and the next four
            const f4v a = *(const f4v*)(lrow + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; i++) win[r][4 * q + i] = a[i];
        }
    };
#pragma unroll
    for (int r = 0; r < 5; r++) read_win(r);
    float sum_g = 0.f;
    unsigned amx = 0u;          // fp32 tensors: running max |stored value| (one VALU operation per output beside 16 multiply-adds; the pass is HBM-bound)
    // gain -> [EPI 1: * scale + bias -> activation -> gain -> clamp] / [EPI 3: activation gradient at the forward output yref]
    auto finish = [&](float fir, float sc, float bi, float yref) {
        float t = fir * p.gain;
        if constexpr (EPI == 1) {   // bias -> activation -> gain -> clamp, the operation order of bias_act.cu:51-142 (grad 0); identical to the lanes kernel's
            t = t * sc;
            t = t + bi;
            if (p.ep_act == 3) t = (t > 0.f) ? t : t * p.ep_alpha;
            t *= p.ep_gain;
            if (p.ep_clamp >= 0.f) t = (t > -p.ep_clamp & t < p.ep_clamp) ? t : (t >= 0.f) ? p.ep_clamp : -p.ep_clamp;
        }
        if constexpr (EPI == 3) {   // grad-1 form (bias_act.cu:60-61,133-142): dy * act'(pre) * gain, zero where the forward output was clamped
            const float yy = yref * inv_gain;
            float g = t;
            if (p.ep_act == 3) g = (yy > 0.f) ? t : t * p.ep_alpha;
            g *= p.ep_gain;
            if (p.ep_clamp >= 0.f) g = (yref > -p.ep_clamp & yref < p.ep_clamp) ? g : 0.f;
            t = g;
        }
        return t;
    };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int oy = oy0 + 4 * wave + k;
        if (oy >= p.out_h) break;   // wave-uniform
        if (k == 2) { read_win(5); read_win(6); }
        float o[NOUT];
        // The 16 multiply-adds of an output, for two neighbouring outputs at a time: v_pk_fma_f32 (two fp32 FMAs per lane and instruction; hipcc keeps a
        // copy of the window row shifted by one column for the odd taps).  Every output still sees its own chain in the reference's tap order -- the
        // results are bit-identical to the scalar form -- at half the FMA issue slots: on 16-bit tensors the pass is bound by its VALU work, not by HBM
        // (profiles/r03_ufd_tile_valu_ablation.log: bf16 2.9 TB/s, 3.35 without the multiply-adds, fp32 unchanged at 5.5).
        // (out_w = 4 k + 1: EVERY lane carries a fifth output and the last lane of a row stores it next to its four.  Round 6 tried the column as a phase of its own
        // -- one thread per (row, plane), 16 multiply-adds once per workgroup instead of 25 % more in the whole loop: 6 % (256 -> 257) to 31 % (128 -> 129) SLOWER,
        // profiles/r06_c12_*: the lone dword then reaches memory long after the rest of its cache line.)
        float fir[NOUT];
#if defined(SGV_TILE_ABL) && (SGV_TILE_ABL & 1)     // lab build (tools/gpu_recipes): the pass without its 16 multiply-adds per output
#pragma unroll
        for (int v = 0; v < NOUT; v++) fir[v] = win[k + 1][v + 1] * ff[1][1];
#elif (SGV_TILE_LAB_OFF & 16)                      // lab build: one scalar chain per output instead of v_pk_fma_f32 pairs
#pragma unroll
        for (int v = 0; v < NOUT; v++) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc = __builtin_fmaf(win[k + j][v + i], ff[j][i], acc);
            fir[v] = acc;
        }
#else
        typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int v2 = 0; v2 < NOUT / 2; v2++) {
            f2v acc2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc2 = __builtin_elementwise_fma(f2v{win[k + j][2 * v2 + i], win[k + j][2 * v2 + i + 1]}, f2v{ff[j][i], ff[j][i]}, acc2);
            fir[2 * v2] = acc2[0];
            fir[2 * v2 + 1] = acc2[1];
        }
        if constexpr (NOUT & 1) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc = __builtin_fmaf(win[k + j][NOUT - 1 + i], ff[j][i], acc);
            fir[NOUT - 1] = acc;
        }
#endif
#pragma unroll
        for (int v = 0; v < NOUT; v++) {
            const bool stored = v < CPL ? st_vec : st_xtra;
            const float t = finish(fir[v], ep_sc, ep_bi, EPI == 3 ? yo[k][v] : 0.f);
            if constexpr (EPI == 3) { if (stored) sum_g += t; }
            o[v] = t;
            if constexpr (sizeof(T) == 4 && !(SGV_TILE_LAB_OFF & 32)) { if (stored) amx = sgv_amax_fold(amx, t); }
        }
        T* yr = yp + (size_t)oy * p.out_w + ox;
        if (st_vec) { if (NT) store_vec_nt<T, CPL>(yr, o); else store_vec_plain<T, CPL>(yr, o); }
        if constexpr (XTRA) { if (st_xtra) sgv_traits<T>::store(yr + CPL, o[CPL]); }
    }
    if constexpr (EPI == 3) {   // reduce over the lanes of a plane row, then one atomic per plane and wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
            if (off < lpr) sum_g += __shfl_xor(sum_g, off, 64);
        if (sub == 0 && plane_ok && oy0 + 4 * wave < p.out_h) atomicAdd(p.ep_sum_g + plane, sum_g);
    }
    if constexpr (sizeof(T) == 4) { if (p.y_amax) sgv_amax_commit_wg(amx, p.y_amax, (unsigned*)(tile_lds + p.lds_amax_word)); }
}

// ---------------------------------------------------------------------------------------------
// 2x down-sampling tile kernel (up = 1, down = 2 on both axes, filter <= 4x4, pad 0..3): the discriminator's skip branch and the gradient of a 2x up-sampling.
// Round 6.  The lane-exchange kernel walks a strip per wave -- loads, arithmetic and stores interleaved for the wave's lifetime -- and runs these calls at
// 4.0-4.7 TB/s; the LDS-tile structure of upfirdn2d_tile_kernel (every load of the workgroup in flight at t = 0, one barrier, short-lived workgroups, the row
// tiles of a plane on one XCD) reaches 6.0-6.3 on the FIR passes.  Same structure here: a workgroup owns 8 output rows of 64 >> lpr_log2 planes; the 18 input
// rows of each are dealt to the lanes as 16-byte requests (36 wave rows, 9 per wave), parked zero-padded in LDS (position q of a plane row = input column
// q - pad_x); wave w then produces output rows 2 w and 2 w + 1 of every plane from LDS rows 4 w .. 4 w + 5: three aligned ds_read_b128 per lane and row (input
// columns 8 l - pad_x .. + 11 for output columns 4 l .. 4 l + 3), one fmaf chain per output in the reference's tap order (ascending input row, then column).
struct down2_params {
    const void* x;
    const float* f;
    void* y;
    int flip;
    float gain;
    int in_w, in_h, out_w, out_h;
    int planes;
    int f_w, f_h;
    int64_t f_sw, f_sh;
    int pad_x, pad_y;
    int lpr_log2;      // lanes per OUTPUT row of a plane (out_w <= 4 << lpr_log2, 1 .. 5); 64 >> lpr_log2 planes side by side in a wave
    int row_tiles;     // ceil(out_h / 8)
    int xcd_blocks;    // as tile_params
    int lds_amax_word; // as tile_params
    float* y_amax;
};
constexpr int D2_ROWS = 8, D2_IN_ROWS = 2 * D2_ROWS + 2;
inline int down2_lds_floats(int lpr_log2) { return D2_IN_ROWS * (64 >> lpr_log2) * ((8 << lpr_log2) + 8); }

template <typename T, bool NT, bool F44>
__global__ __launch_bounds__(256) void upfirdn2d_down2_tile_kernel(down2_params p) {
    extern __shared__ __attribute__((aligned(16))) float tile_lds[];
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rt, pg;
    if ((int)blockIdx.x < p.xcd_blocks) { const int j = blockIdx.x >> 3; rt = j % p.row_tiles; pg = (j / p.row_tiles) * 8 + (blockIdx.x & 7); }
    else { rt = blockIdx.x % p.row_tiles; pg = blockIdx.x / p.row_tiles; }
    const int lpr = 1 << p.lpr_log2, ppw = 64 >> p.lpr_log2;
    const int seg_pitch = 8 * lpr + 8, row_pitch = ppw * seg_pitch;
    float ff[4][4];
    if constexpr (F44) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) ff[a][b] = p.flip ? p.f[a * 4 + b] : p.f[(3 - a) * 4 + (3 - b)];
    }
    const int oy0 = rt * D2_ROWS;
    const int iy0 = 2 * oy0 - p.pad_y;

    // ---- load phase: 18 rows x ppw planes = 36 wave rows of 64 lanes x 4 columns; wave row g = wave + 4 k holds the segments g * spw .. (rows of a plane fastest) ----
    const int nl_log2 = p.lpr_log2 + 1, nl = 2 * lpr;            // lanes per INPUT row of a plane
    const int sub = lane & (nl - 1), seg_in_row = lane >> nl_log2, spw = 64 >> nl_log2;
    const int ix0 = 4 * sub - p.pad_x;
    const int base = min(max(ix0, 0), p.in_w - 4);
    const int sh = base - ix0;
    const int over_u = (((-p.pad_x - p.in_w) % 4) + 4) % 4;
    const bool cols_dead = ix0 >= p.in_w || ix0 + 3 < 0;
    const int ixh = 4 * nl - p.pad_x + sub;                       // lanes sub < 8: the columns behind the last lane's block
    const bool halo_ok = sub < 8 && ixh >= 0 && ixh < p.in_w;
    const int ixh_c = min(max(ixh, 0), p.in_w - 1);
    constexpr int NK = D2_IN_ROWS / 2;                            // 36 wave rows / 4 waves
    float m[NK][4], h[NK];
    int lofs[NK];
    bool rok[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int id = (wave + 4 * k) * spw + seg_in_row;
        const int slot = id / D2_IN_ROWS, r = id - slot * D2_IN_ROWS;
        const int plane = pg * ppw + slot;
        const bool plane_ok = plane < p.planes;
        const int iy = iy0 + r;
        rok[k] = plane_ok && iy >= 0 && iy < p.in_h;
        lofs[k] = r * row_pitch + slot * seg_pitch;
        const T* row = (const T*)p.x + ((size_t)(plane_ok ? plane : 0) * p.in_h + min(max(iy, 0), p.in_h - 1)) * p.in_w;
        if (r >= 2 && r < D2_IN_ROWS - 2) row_loader<T, 4, true>::run(row + base, m[k]);     // rows no other tile reads
        else row_loader<T, 4>::run(row + base, m[k]);
        h[k] = 0.f;
        if (sub < 8) h[k] = sgv_traits<T>::load(row + ixh_c);
    }
    if constexpr (!F44) {
        const int fsh = (int)p.f_sh, fsw = (int)p.f_sw;
        const int a0 = p.flip ? 0 : (p.f_h - 1) * fsh, da = p.flip ? fsh : -fsh;
        const int b0 = p.flip ? 0 : (p.f_w - 1) * fsw, db = p.flip ? fsw : -fsw;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const bool live = a < p.f_h && b < p.f_w;
                const float v = p.f[live ? a0 + a * da + b0 + b * db : 0];
                ff[a][b] = live ? v : 0.f;
            }
    }
#pragma unroll
    for (int k = 0; k < NK; k++) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = m[k][i];
#pragma unroll
        for (int d = 1; d <= 3; d++)
            if (p.pad_x == d) {
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = (sh > 0) ? (i - d >= 0 ? m[k][i - d] : 0.f) : o[i];
            }
#pragma unroll
        for (int d = 1; d < 4; d++)
            if (over_u == d) {
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = (sh < 0) ? (i + d < 4 ? m[k][i + d] : 0.f) : o[i];
            }
        const bool live = rok[k] && !cols_dead;
        float* lrow = tile_lds + lofs[k];
        *(f4v*)(lrow + 4 * sub) = f4v{live ? o[0] : 0.f, live ? o[1] : 0.f, live ? o[2] : 0.f, live ? o[3] : 0.f};
        if (sub < 8) lrow[4 * nl + sub] = (rok[k] && halo_ok) ? h[k] : 0.f;
    }
    if (threadIdx.x == 0) tile_lds[p.lds_amax_word] = 0.f;
    __syncthreads();

    // ---- compute phase: wave w -> output rows 2 w, 2 w + 1 of every plane, from LDS rows 4 w .. 4 w + 5 ----
    const int l = lane & (lpr - 1), cslot = lane >> p.lpr_log2;
    const int cplane = pg * ppw + cslot;
    const bool cplane_ok = cplane < p.planes;
    const int ox = 4 * l;
    T* yp = (T*)p.y + (size_t)(cplane_ok ? cplane : 0) * p.out_h * p.out_w;
    float win[6][12];
    auto read_win = [&](int r) {
        const float* lrow = tile_lds + (4 * wave + r) * row_pitch + cslot * seg_pitch + 8 * l;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const f4v a = *(const f4v*)(lrow + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; i++) win[r][4 * q + i] = a[i];
        }
    };
#pragma unroll
    for (int r = 0; r < 4; r++) read_win(r);
    unsigned amx = 0u;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int oy = oy0 + 2 * wave + u;
        if (oy >= p.out_h) break;       // wave-uniform
        if (u == 1) { read_win(4); read_win(5); }
        float o[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ky++)
#pragma unroll
                for (int kx = 0; kx < 4; kx++) acc = __builtin_fmaf(win[2 * u + ky][2 * v + kx], ff[ky][kx], acc);
            o[v] = acc * p.gain;
            if constexpr (sizeof(T) == 4) { if (cplane_ok && ox + v < p.out_w) amx = sgv_amax_fold(amx, o[v]); }
        }
        T* yr = yp + (size_t)oy * p.out_w + ox;
        if (cplane_ok && ox + 4 <= p.out_w) { if (NT) store_vec_nt<T, 4>(yr, o); else store_vec_plain<T, 4>(yr, o); }
        else if (cplane_ok) {
#pragma unroll
            for (int v = 0; v < 3; v++) if (ox + v < p.out_w) sgv_traits<T>::store(yr + v, o[v]);
        }
    }
    if constexpr (sizeof(T) == 4) { if (p.y_amax) sgv_amax_commit_wg(amx, p.y_amax, (unsigned*)(tile_lds + p.lds_amax_word)); }
}

// ---------------------------------------------------------------------------------------------
// 2x up-sampling tile kernel (up = 2, down = 1 on both axes, filter <= 4x4, pad 0..3): the skip-RGB up-sampling and the gradient of a 2x down-sampling
// (ADD: + the gradient that arrived from the tensor's other consumer, sgv_upfirdn2d_fused mode 4).  Round 6, same structure as the kernels above.  The pass
// writes four times what it reads: a workgroup owns 16 output rows x (64 >> lpr_log2) planes x 4 << lpr_log2 columns; their <= 10 input rows are five wave rows
// of 16-byte requests, parked zero-padded in LDS (position q of a plane row = input column q + floor((1 - pad_x) / 2)); wave w produces output rows 4 w .. 4 w + 3
// from four LDS rows, each output from its 2 x 2 live taps (a zero-inserted image never multiplies the other 12) in the reference's order (input row, then column).
struct up2_params {
    const void* x;
    const float* f;
    void* y;
    const void* addend;  // ADD: OUTPUT-shaped summand, dtype / layout of y
    int flip;
    float gain;
    int in_w, in_h, out_w, out_h;
    int planes;
    int f_w, f_h;
    int64_t f_sw, f_sh;
    int pad_x, pad_y;
    int lpr_log2;      // lanes per output row of a plane (out_w <= 4 << lpr_log2, 2 .. 6)
    int row_tiles;     // ceil(out_h / 16)
    int xcd_blocks;
    int lds_amax_word;
    float* y_amax;
};
constexpr int U2_ROWS = 16, U2_IN_ROWS = 10;
inline int up2_lds_floats(int lpr_log2) { return U2_IN_ROWS * (64 >> lpr_log2) * ((2 << lpr_log2) + 4); }

template <typename T, bool NT, bool ADD>
__global__ __launch_bounds__(256) void upfirdn2d_up2_tile_kernel(up2_params p) {
    extern __shared__ __attribute__((aligned(16))) float tile_lds[];
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rt, pg;
    if ((int)blockIdx.x < p.xcd_blocks) { const int j = blockIdx.x >> 3; rt = j % p.row_tiles; pg = (j / p.row_tiles) * 8 + (blockIdx.x & 7); }
    else { rt = blockIdx.x % p.row_tiles; pg = blockIdx.x / p.row_tiles; }
    const int lpr = 1 << p.lpr_log2, ppw = 64 >> p.lpr_log2;
    const int seg_pitch = 2 * lpr + 4, row_pitch = ppw * seg_pitch;
    const int tx = 1 - p.pad_x, ty = 1 - p.pad_y;                 // mid = o + t; input index = floor(mid / 2); first live tap = 1 - (mid & 1)
    const int cb = tx >> 1, rb = ty >> 1;                          // floor(t / 2): -1 or 0
    const int oy0 = rt * U2_ROWS;
    const int iyb = (oy0 >> 1) + rb;                               // input row of LDS row 0 (oy0 is even)

    // ---- load phase: 10 rows x ppw planes, lpr / 2 lanes x 4 columns per row: five wave rows ----
    const int nl_log2 = p.lpr_log2 - 1, nl = lpr >> 1;
    const int sub = lane & (nl - 1), seg_in_row = lane >> nl_log2, spw = 64 >> nl_log2;
    const int ix0 = 4 * sub + cb;
    const int base = min(max(ix0, 0), p.in_w - 4);
    const int sh = base - ix0;
    const int over_u = (((cb - p.in_w) % 4) + 4) % 4;
    const bool cols_dead = ix0 >= p.in_w || ix0 + 3 < 0;
    const int ixh = 4 * nl + cb + sub;
    const bool halo_ok = sub < 4 && ixh >= 0 && ixh < p.in_w;
    const int ixh_c = min(max(ixh, 0), p.in_w - 1);
    float m[2][4], h[2];
    int lofs[2];
    bool rok[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int g = wave + 4 * k;
        const int id = min(g, 4) * spw + seg_in_row;
        const int slot = id / U2_IN_ROWS, r = id - slot * U2_IN_ROWS;
        const int plane = pg * ppw + slot;
        const bool plane_ok = plane < p.planes;
        const int iy = iyb + r;
        rok[k] = plane_ok && iy >= 0 && iy < p.in_h;
        lofs[k] = r * row_pitch + slot * seg_pitch;
#pragma unroll
        for (int i = 0; i < 4; i++) m[k][i] = 0.f;
        h[k] = 0.f;
        if (g < 5) {        // wave-uniform
            const T* row = (const T*)p.x + ((size_t)(plane_ok ? plane : 0) * p.in_h + min(max(iy, 0), p.in_h - 1)) * p.in_w;
            row_loader<T, 4>::run(row + base, m[k]);
            if (sub < 4) h[k] = sgv_traits<T>::load(row + ixh_c);
        }
    }
    // taps: ff[a][b] = flipped, zero-padded filter (the convention of every kernel in this file), through the scalar cache behind the row loads
    float ff[4][4];
    {
        const int fsh = (int)p.f_sh, fsw = (int)p.f_sw;
        const int a0 = p.flip ? 0 : (p.f_h - 1) * fsh, da = p.flip ? fsh : -fsh;
        const int b0 = p.flip ? 0 : (p.f_w - 1) * fsw, db = p.flip ? fsw : -fsw;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const bool live = a < p.f_h && b < p.f_w;
                const float v = p.f[live ? a0 + a * da + b0 + b * db : 0];
                ff[a][b] = live ? v : 0.f;
            }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (wave + 4 * k >= 5) continue;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = m[k][i];
        if (cb == -1) {
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = (sh > 0) ? (i >= 1 ? m[k][i - 1] : 0.f) : o[i];
        }
#pragma unroll
        for (int d = 1; d < 4; d++)
            if (over_u == d) {
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = (sh < 0) ? (i + d < 4 ? m[k][i + d] : 0.f) : o[i];
            }
        const bool live = rok[k] && !cols_dead;
        float* lrow = tile_lds + lofs[k];
        *(f4v*)(lrow + 4 * sub) = f4v{live ? o[0] : 0.f, live ? o[1] : 0.f, live ? o[2] : 0.f, live ? o[3] : 0.f};
        if (sub < 4) lrow[4 * nl + sub] = (rok[k] && halo_ok) ? h[k] : 0.f;
    }
    if (threadIdx.x == 0) tile_lds[p.lds_amax_word] = 0.f;
    __syncthreads();

    // ---- compute phase: wave w -> output rows 4 w .. 4 w + 3 from LDS rows rw .. rw + 3 ----
    const int l = lane & (lpr - 1), cslot = lane >> p.lpr_log2;
    const int cplane = pg * ppw + cslot;
    const bool cplane_ok = cplane < p.planes;
    const int ox = 4 * l;
    const int rw = ((4 * wave + ty) >> 1) - rb;
    float win[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float* lrow = tile_lds + min(rw + r, U2_IN_ROWS - 1) * row_pitch + cslot * seg_pitch + 2 * l;
        const f2v a = *(const f2v*)lrow, b = *(const f2v*)(lrow + 2);
        win[r][0] = a[0]; win[r][1] = a[1]; win[r][2] = b[0]; win[r][3] = b[1];
    }
    T* yp = (T*)p.y + (size_t)(cplane_ok ? cplane : 0) * p.out_h * p.out_w;
    const T* ap = ADD ? (const T*)p.addend + (size_t)(cplane_ok ? cplane : 0) * p.out_h * p.out_w : nullptr;
    const bool st_vec = cplane_ok && ox + 4 <= p.out_w, st_any = cplane_ok && ox < p.out_w;
    unsigned amx = 0u;
    // one output: rows dr, dr + 1 and columns dc, dc + 1 of the window, taps (a0, a0 + 2) x (b0, b0 + 2)
    auto out1 = [&](int dr, int a0, int dc, int b0) {
        float acc = __builtin_fmaf(win[dr][dc], ff[a0][b0], 0.f);
        acc = __builtin_fmaf(win[dr][dc + 1], ff[a0][b0 + 2], acc);
        acc = __builtin_fmaf(win[dr + 1][dc], ff[a0 + 2][b0], acc);
        acc = __builtin_fmaf(win[dr + 1][dc + 1], ff[a0 + 2][b0 + 2], acc);
        return acc * p.gain;
    };
    auto row_out = [&](int k, int dr, int a0) {
        const int oy = oy0 + 4 * wave + k;
        if (oy >= p.out_h) return;      // wave-uniform
        float o[4];
        if ((tx & 1) == 0) { o[0] = out1(dr, a0, 0, 1); o[1] = out1(dr, a0, 0, 0); o[2] = out1(dr, a0, 1, 1); o[3] = out1(dr, a0, 1, 0); }
        else               { o[0] = out1(dr, a0, 0, 0); o[1] = out1(dr, a0, 1, 1); o[2] = out1(dr, a0, 1, 0); o[3] = out1(dr, a0, 2, 1); }
        T* yr = yp + (size_t)oy * p.out_w + ox;
        if constexpr (ADD) {
            float yo[4] = {0.f, 0.f, 0.f, 0.f};
            const T* ar = ap + (size_t)oy * p.out_w + ox;
            if (st_vec) row_loader<T, 4>::run(ar, yo);
            else if (st_any) {
#pragma unroll
                for (int v = 0; v < 3; v++) if (ox + v < p.out_w) yo[v] = sgv_traits<T>::load(ar + v);
            }
#pragma unroll
            for (int v = 0; v < 4; v++) o[v] += yo[v];
        }
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int v = 0; v < 4; v++) if (cplane_ok && ox + v < p.out_w) amx = sgv_amax_fold(amx, o[v]);
        }
        if (st_vec) { if (NT) store_vec_nt<T, 4>(yr, o); else store_vec_plain<T, 4>(yr, o); }
        else if (st_any) {
#pragma unroll
            for (int v = 0; v < 3; v++) if (ox + v < p.out_w) sgv_traits<T>::store(yr + v, o[v]);
        }
    };
    if ((ty & 1) == 0) { row_out(0, 0, 1); row_out(1, 0, 0); row_out(2, 1, 1); row_out(3, 1, 0); }
    else               { row_out(0, 0, 0); row_out(1, 1, 1); row_out(2, 1, 0); row_out(3, 2, 1); }
    if constexpr (sizeof(T) == 4) { if (p.y_amax) sgv_amax_commit_wg(amx, p.y_amax, (unsigned*)(tile_lds + p.lds_amax_word)); }
}

typedef void (*lanes_fn)(lanes_params);
constexpr int LANES_WPB = 4;  // waves per workgroup (8 measured equal: the halo hand-off already covers 3 of 4 strip seams)

template <typename T>
lanes_fn pick_lanes_kernel(const sgv_upfirdn2d_params* p, int xtra, bool seg, int epi) {
    const int u = p->up_x, d = p->down_x, px = p->pad_x0, py = p->pad_y0;
    if (p->up_y != u || p->down_y != d || p->f_w > 4 || p->f_h > 4) return nullptr;
#define SGV_LANES_EPI(U, D, PX, PY, E)                                                                \
    if (u == U && d == D && px == PX && py == PY && epi == E)                                            \
        return seg ? (xtra ? (lanes_fn)upfirdn2d_lanes_kernel<T, U, D, PX, PY, 1, true, LANES_WPB, E> : (lanes_fn)upfirdn2d_lanes_kernel<T, U, D, PX, PY, 0, true, LANES_WPB, E>)         \
                   : (xtra ? (lanes_fn)upfirdn2d_lanes_kernel<T, U, D, PX, PY, 1, false, LANES_WPB, E> : (lanes_fn)upfirdn2d_lanes_kernel<T, U, D, PX, PY, 0, false, LANES_WPB, E>);
    SGV_LANES_EPI(1, 1, 1, 1, 1)   // synthesis-layer epilogue: FIR (2r+1 -> 2r) * dcoefs + bias -> lrelu -> clamp
    SGV_LANES_EPI(1, 1, 2, 2, 2)   // its backward: lrelu'/clamp mask * dcoefs -> FIR (2r -> 2r+1), plane sums
    SGV_LANES_EPI(1, 1, 1, 1, 3)   // backward of "activation, then the FIR in front of a strided convolution": FIR (2r+1 -> 2r), then lrelu'/clamp mask
    SGV_LANES_EPI(2, 1, 2, 2, 4)   // backward of the 2x down-sampling FIR (= 2x up-sampling) + the gradient from the input's other consumer
#undef SGV_LANES_EPI
    if (epi != 0) return nullptr;
#define SGV_LANES(U, D, PX, PY)                                                                     \
    if (u == U && d == D && px == PX && py == PY)                                                   \
        return seg ? (xtra ? (lanes_fn)upfirdn2d_lanes_kernel<T, U, D, PX, PY, 1, true, LANES_WPB, 0> : (lanes_fn)upfirdn2d_lanes_kernel<T, U, D, PX, PY, 0, true, LANES_WPB, 0>)         \
                   : (xtra ? (lanes_fn)upfirdn2d_lanes_kernel<T, U, D, PX, PY, 1, false, LANES_WPB, 0> : (lanes_fn)upfirdn2d_lanes_kernel<T, U, D, PX, PY, 0, false, LANES_WPB, 0>);
    SGV_LANES(1, 1, 1, 1)   // FIR after the up-convolution (2r+1 -> 2r); backward of the D pre-FIR
    SGV_LANES(1, 1, 2, 2)   // FIR before the strided convolution (r -> r+1); backward of the G FIR
    SGV_LANES(2, 1, 2, 2)   // 2x upsample (skip-RGB); backward of the 2x downsample
    SGV_LANES(1, 2, 1, 1)   // 2x downsample (D skip); backward of the 2x upsample
#undef SGV_LANES
    return nullptr;
}

struct lanes_plan {
    lanes_fn fn;
    lanes_params lp;
    int blocks;
    int threads;
};


inline int pymod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }

typedef void (*rows_fn)(rows_params);

// Instantiation table: (up, down) per axis in {(1,1),(2,1),(1,2)}; 4x4 padded filter for 2-D, and
// the phase R0 = (up-1-pad0) mod up where up == 2.
template <typename T, int UPX, int UPY, int DOWNX, int DOWNY, int FWP, int FHP>
rows_fn pick_phase(int r0x, int r0y, int xtra) {
#define SGV_PICK(RX, RY)                                                                                  \
    if (r0x == RX && r0y == RY)                                                                           \
        return xtra ? (rows_fn)upfirdn2d_rows_kernel<T, UPX, UPY, DOWNX, DOWNY, FWP, FHP, RX, RY, 1>      \
                    : (rows_fn)upfirdn2d_rows_kernel<T, UPX, UPY, DOWNX, DOWNY, FWP, FHP, RX, RY, 0>;
    SGV_PICK(0, 0)
    if constexpr (UPX == 2) { SGV_PICK(1, 0) }
    if constexpr (UPY == 2) { SGV_PICK(0, 1) }
    if constexpr (UPX == 2 && UPY == 2) { SGV_PICK(1, 1) }
#undef SGV_PICK
    return nullptr;
}

template <typename T>
rows_fn pick_rows_kernel(const sgv_upfirdn2d_params* p, int r0x, int r0y, int xtra) {
    const int ux = p->up_x, uy = p->up_y, dx = p->down_x, dy = p->down_y;
    if (p->f_w <= 4 && p->f_h <= 4) {
        if (ux == 1 && uy == 1 && dx == 1 && dy == 1) return pick_phase<T, 1, 1, 1, 1, 4, 4>(r0x, r0y, xtra);
        if (ux == 2 && uy == 2 && dx == 1 && dy == 1) return pick_phase<T, 2, 2, 1, 1, 4, 4>(r0x, r0y, xtra);
        if (ux == 1 && uy == 1 && dx == 2 && dy == 2) return pick_phase<T, 1, 1, 2, 2, 4, 4>(r0x, r0y, xtra);
    }
    // One-dimensional passes of separable filters up to 12 taps: the ADA augmentation pipe resamples with the 12-tap
    // 'sym6' wavelet as a horizontal then a vertical upfirdn2d (src/training/augment.py:289,300; the reference serves
    // them with its 1x12 / 12x1 / 1x24 tile kernels, upfirdn2d.cu:250-339).
    if (p->f_h == 1 && p->f_w <= 12 && uy == 1 && dy == 1) {
        if (ux == 1 && dx == 1) return pick_phase<T, 1, 1, 1, 1, 12, 1>(r0x, r0y, xtra);
        if (ux == 2 && dx == 1) return pick_phase<T, 2, 1, 1, 1, 12, 1>(r0x, r0y, xtra);
        if (ux == 1 && dx == 2) return pick_phase<T, 1, 1, 2, 1, 12, 1>(r0x, r0y, xtra);
    }
    if (p->f_w == 1 && p->f_h <= 12 && ux == 1 && dx == 1) {
        if (uy == 1 && dy == 1) return pick_phase<T, 1, 1, 1, 1, 1, 12>(r0x, r0y, xtra);
        if (uy == 2 && dy == 1) return pick_phase<T, 1, 2, 1, 1, 1, 12>(r0x, r0y, xtra);
        if (uy == 1 && dy == 2) return pick_phase<T, 1, 1, 1, 2, 1, 12>(r0x, r0y, xtra);
    }
    return nullptr;
}

bool dense_nchw(int w, int h, int c, int64_t sw, int64_t sh, int64_t sc, int64_t sn) {
    return sw == 1 && sh == w && sc == (int64_t)w * h && sn == (int64_t)w * h * c;
}

struct rows_plan {
    rows_fn fn;
    rows_params rp;
    int blocks;
};

bool plan_rows(const sgv_upfirdn2d_params* p, int dtype, rows_plan* plan) {
    if (dtype == SGV_F64) return false;
    if (!dense_nchw(p->in_w, p->in_h, p->in_c, p->in_sw, p->in_sh, p->in_sc, p->in_sn)) return false;
    if (!dense_nchw(p->out_w, p->out_h, p->in_c, p->out_sw, p->out_sh, p->out_sc, p->out_sn)) return false;
    const int r0x = pymod(p->up_x - 1 - p->pad_x0, p->up_x);
    const int r0y = pymod(p->up_y - 1 - p->pad_y0, p->up_y);
    const int xtra = (p->out_w > VEC && p->out_w % VEC == 1) ? 1 : 0;
    rows_fn fn = nullptr;
    if (dtype == SGV_F32) fn = pick_rows_kernel<float>(p, r0x, r0y, xtra);
    if (dtype == SGV_F16) fn = pick_rows_kernel<sgv_half_t>(p, r0x, r0y, xtra);
    if (dtype == SGV_BF16) fn = pick_rows_kernel<sgv_bf16_t>(p, r0x, r0y, xtra);
    if (!fn) return false;

    rows_params& rp = plan->rp;
    rp.x = p->x; rp.f = p->f; rp.y = p->y;
    rp.pad_x0 = p->pad_x0; rp.pad_y0 = p->pad_y0; rp.flip = p->flip; rp.gain = p->gain;
    rp.in_w = p->in_w; rp.in_h = p->in_h; rp.out_w = p->out_w; rp.out_h = p->out_h;
    rp.planes = p->in_c * p->in_n;
    rp.f_w = p->f_w; rp.f_h = p->f_h; rp.f_sw = p->f_sw; rp.f_sh = p->f_sh;
    rp.xtra = xtra;
    const int n_main = xtra ? p->out_w - 1 : p->out_w;
    const int cbs = (n_main + VEC - 1) / VEC;
    int lpr_log2 = 0;
    while ((1 << lpr_log2) < cbs && lpr_log2 < 6) lpr_log2++;
    rp.lpr_log2 = lpr_log2;
    rp.col_groups = (cbs + (1 << lpr_log2) - 1) >> lpr_log2;
    const int ppw = 64 >> lpr_log2;
    rp.plane_groups = (rp.planes + ppw - 1) / ppw;
    // Strip height: aim for >= ~16K waves so all 1024 SIMDs hold several waves, keep strips >= 8 rows
    // (halo overhead (FH-1)/strip_h) and a multiple of the row group G = up_y.
    const int g = p->up_y;
    int64_t base_waves = (int64_t)rp.plane_groups * rp.col_groups;
    int strip_h = 32;
    while (strip_h > 8 && base_waves * ((p->out_h + strip_h - 1) / strip_h) < 16384) strip_h >>= 1;
    if (strip_h > p->out_h) strip_h = p->out_h;
    strip_h = ((strip_h + g - 1) / g) * g;
    rp.strip_h = strip_h;
    rp.strips = (p->out_h + strip_h - 1) / strip_h;
    int64_t waves = base_waves * rp.strips;
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffff) return false;
    plan->fn = fn;
    plan->blocks = (int)blocks;
    return true;
}


bool plan_lanes(const sgv_upfirdn2d_params* p, int dtype, lanes_plan* plan, const sgv_fir_epilogue* epi = nullptr) {
    if (dtype == SGV_F64) return false;
    if (!dense_nchw(p->in_w, p->in_h, p->in_c, p->in_sw, p->in_sh, p->in_sc, p->in_sn)) return false;
    if (!dense_nchw(p->out_w, p->out_h, p->in_c, p->out_sw, p->out_sh, p->out_sc, p->out_sn)) return false;
    const int xtra = (p->out_w > VEC && p->out_w % VEC == 1) ? 1 : 0;
    const int n_main = xtra ? p->out_w - 1 : p->out_w;
    const int cbs = (n_main + VEC - 1) / VEC;
    const bool seg = cbs <= 32;  // narrow rows: several planes per wave, segmented lane exchange
    lanes_fn fn = nullptr;
    constexpr int wpb = LANES_WPB;
    const int emode = epi ? epi->mode : 0;
    if (dtype == SGV_F32) fn = pick_lanes_kernel<float>(p, xtra, seg, emode);
    if (dtype == SGV_F16) fn = pick_lanes_kernel<sgv_half_t>(p, xtra, seg, emode);
    if (dtype == SGV_BF16) fn = pick_lanes_kernel<sgv_bf16_t>(p, xtra, seg, emode);
    if (!fn) return false;
    static const int strip_env = []() { const char* e = getenv("SGV_LANES_STRIP"); return e ? atoi(e) : 0; }();
    lanes_params& lp = plan->lp;
    lp.x = p->x; lp.f = p->f; lp.y = p->y; lp.flip = p->flip; lp.gain = p->gain;
    lp.in_w = p->in_w; lp.in_h = p->in_h; lp.out_w = p->out_w; lp.out_h = p->out_h;
    lp.planes = p->in_c * p->in_n;
    lp.f_w = p->f_w; lp.f_h = p->f_h; lp.f_sw = p->f_sw; lp.f_sh = p->f_sh;
    lp.col_groups = seg ? 1 : (cbs + 63) / 64;
    int lpr_log2 = 0;
    while ((1 << lpr_log2) < cbs) lpr_log2++;
    lp.lpr_log2 = seg ? lpr_log2 : 6;
    lp.plane_groups = seg ? (lp.planes + (64 >> lpr_log2) - 1) / (64 >> lpr_log2) : lp.planes;
    const double out_bytes = (double)p->out_w * p->out_h * lp.planes * sgv_dtype_size(dtype);
    lp.nt_store = out_bytes > 300e6 ? 1 : 0;
    static const int lds_env = []() { const char* e = getenv("SGV_LANES_LDS"); return e ? atoi(e) : 1; }();
    lp.lds_share = lds_env;
    lp.ep_scale = nullptr; lp.ep_bias = nullptr; lp.ep_yref = nullptr; lp.ep_sum_g = nullptr; lp.ep_sum_gv = nullptr;
    lp.ep_act = 1; lp.ep_alpha = 0.f; lp.ep_gain = 1.f; lp.ep_clamp = -1.f; lp.chans = p->in_c; lp.y_amax = nullptr;
    if (epi) {
        lp.ep_scale = epi->scale; lp.ep_bias = epi->bias; lp.ep_yref = epi->yref; lp.ep_sum_g = epi->sum_g; lp.ep_sum_gv = epi->sum_gv;
        lp.ep_act = epi->act; lp.ep_alpha = epi->alpha; lp.ep_gain = epi->gain; lp.ep_clamp = epi->clamp;
    }
    // Short strips: many short-lived waves whose concurrent footprint is a compact moving window of memory
    // stream HBM best (16-row strips: 5.4 TB/s, 32-row: 4.9 TB/s, 64-row: 4.6 TB/s on the headline call).
    int strip_h = strip_env > 0 ? strip_env : 16;
    if (strip_h > p->out_h) strip_h = p->out_h;
    strip_h = ((strip_h + p->up_y - 1) / p->up_y) * p->up_y;
    lp.strip_h = strip_h;
    lp.strips = (p->out_h + strip_h - 1) / strip_h;
    const int64_t waves = (int64_t)lp.plane_groups * lp.col_groups * lp.strips;
    const int64_t blocks = (waves + wpb - 1) / wpb;
    if (blocks > 0x7fffffff) return false;
    plan->fn = fn;
    plan->blocks = (int)blocks;
    plan->threads = 64 * wpb;
    return true;
}

// ---- upfirdn2d_tile_kernel: planning and launch ----
bool tile_geometry(const sgv_upfirdn2d_params* p, int dtype, int* lpr_log2, int* col_groups, int* xtra, int* cpl_out = nullptr) {
    static const int tile_on = []() { const char* e = getenv("SGV_UFD_TILE"); return e ? atoi(e) : 1; }();   // SGV_UFD_TILE=0: the strip-walking kernels of rounds 1-2
    if (!tile_on || dtype == SGV_F64) return false;
    if (p->up_x != 1 || p->up_y != 1 || p->down_x != 1 || p->down_y != 1 || p->f_w > 4 || p->f_h > 4) return false;
    if (p->pad_x0 < 0 || p->pad_x0 > 3 || p->pad_y0 < 0 || p->pad_y0 > 3 || p->in_w < 4) return false;
    if (!dense_nchw(p->in_w, p->in_h, p->in_c, p->in_sw, p->in_sh, p->in_sc, p->in_sn)) return false;
    if (!dense_nchw(p->out_w, p->out_h, p->in_c, p->out_sw, p->out_sh, p->out_sc, p->out_sn)) return false;
    const int xt = (p->out_w > 4 && p->out_w % 4 == 1) ? 1 : 0;
    const int n_main = xt ? p->out_w - 1 : p->out_w;
    if (n_main % 4 != 0) return false;
    // 16-bit tensors: eight columns per lane (16-byte requests) where the row divides -- SGV_UFD_TILE_CPL8=1; off by default: measured equal (fused
    // modes) or 6 % slower (plain pass) than four columns in the mixed-precision step (profiles/r03_ufd_tile_cpl8_ab.log): the 16-bit pass runs at the
    // fp32 pass's OUTPUT-ELEMENT rate (0.7 T/s) whatever the request size, and neither halving its FMA instructions (v_pk_fma_f32) nor the select
    // chain moved it (profiles/r03_ufd_tile_pkfma.log)
    static const int cpl8_on = []() { const char* e = getenv("SGV_UFD_TILE_CPL8"); return e ? atoi(e) : 0; }();
    const int cpl = (cpl8_on && cpl_out && (dtype == SGV_F16 || dtype == SGV_BF16) && n_main % 8 == 0 && n_main >= 32 && p->in_w >= 8) ? 8 : 4;
    if (cpl_out) *cpl_out = cpl;
    const int cbs = n_main / cpl;
    if (cbs <= 32) {
        int l = 2;
        while ((1 << l) < cbs) l++;
        *lpr_log2 = l; *col_groups = 1;
    } else {
        *lpr_log2 = 6; *col_groups = (cbs + 63) / 64;
    }
    *xtra = xt;
    return true;
}

template <typename T, int X, int E, int CPL>
void launch_tile_xec(const tile_params& tp, bool f44, dim3 grid, size_t lds, hipStream_t stream) {
    // dense 4 x 4 filters get the specialised forms (WIDE x NT); any other filter the general one
    if (!f44) { hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, X, E, false, false, false, CPL>), grid, dim3(256), lds, stream, tp); return; }
    if (tp.lpr_log2 == 6) {
        if (tp.nt_store) hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, X, E, true, true, true, CPL>), grid, dim3(256), lds, stream, tp);
        else hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, X, E, true, false, true, CPL>), grid, dim3(256), lds, stream, tp);
    } else {
        if (tp.nt_store) hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, X, E, false, true, true, CPL>), grid, dim3(256), lds, stream, tp);
        else hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, X, E, false, false, true, CPL>), grid, dim3(256), lds, stream, tp);
    }
}

template <typename T, int X, int E>
void launch_tile_xe(const tile_params& tp, int cpl, bool f44, dim3 grid, size_t lds, hipStream_t stream) {
    if constexpr (sizeof(T) == 2) { if (cpl == 8) { launch_tile_xec<T, X, E, 8>(tp, f44, grid, lds, stream); return; } }
    launch_tile_xec<T, X, E, 4>(tp, f44, grid, lds, stream);
}

template <typename T>
void launch_tile_t(const tile_params& tp, int xtra, int epi, int cpl, bool f44, dim3 grid, size_t lds, hipStream_t stream) {
    if (epi == 0) { if (xtra) launch_tile_xe<T, 1, 0>(tp, cpl, f44, grid, lds, stream); else launch_tile_xe<T, 0, 0>(tp, cpl, f44, grid, lds, stream); }
    else if (epi == 1) { if (xtra) launch_tile_xe<T, 1, 1>(tp, cpl, f44, grid, lds, stream); else launch_tile_xe<T, 0, 1>(tp, cpl, f44, grid, lds, stream); }
    else if (epi == 2) { if (xtra) launch_tile_xe<T, 1, 2>(tp, cpl, f44, grid, lds, stream); else launch_tile_xe<T, 0, 2>(tp, cpl, f44, grid, lds, stream); }
    else { if (xtra) launch_tile_xe<T, 1, 3>(tp, cpl, f44, grid, lds, stream); else launch_tile_xe<T, 0, 3>(tp, cpl, f44, grid, lds, stream); }
}

int launch_tile(const sgv_upfirdn2d_params* p, const sgv_fir_epilogue* e, int dtype, int lpr_log2, int col_groups, int xtra, int cpl, hipStream_t stream, sgv_launch_scope& scope) {
    tile_params tp{};
    tp.y_amax = dtype == SGV_F32 ? scope.take_amax_sink() : nullptr;
    tp.x = p->x; tp.f = p->f; tp.y = p->y; tp.flip = p->flip; tp.gain = p->gain;
    tp.in_w = p->in_w; tp.in_h = p->in_h; tp.out_w = p->out_w; tp.out_h = p->out_h; tp.planes = p->in_c * p->in_n;
    tp.f_w = p->f_w; tp.f_h = p->f_h; tp.f_sw = p->f_sw; tp.f_sh = p->f_sh;
    tp.pad_x = p->pad_x0; tp.pad_y = p->pad_y0;
    tp.lpr_log2 = lpr_log2; tp.col_groups = col_groups;
    tp.row_tiles = (p->out_h + TILE_ROWS - 1) / TILE_ROWS;
    const double out_bytes = (double)p->out_w * p->out_h * tp.planes * sgv_dtype_size(dtype);
    tp.nt_store = out_bytes > 300e6 ? 1 : 0;
    tp.ep_act = 1; tp.ep_alpha = 0.f; tp.ep_gain = 1.f; tp.ep_clamp = -1.f; tp.chans = p->in_c;
    const int epi = e ? e->mode : 0;
    if (e) {
        tp.ep_scale = e->scale; tp.ep_bias = e->bias; tp.ep_yref = e->yref; tp.ep_sum_g = e->sum_g; tp.ep_sum_gv = e->sum_gv;
        tp.ep_act = e->act; tp.ep_alpha = e->alpha; tp.ep_gain = e->gain; tp.ep_clamp = e->clamp;
    }
    const int ppw = 64 >> lpr_log2;
    const int64_t units = (int64_t)((tp.planes + ppw - 1) / ppw) * col_groups;
    const int64_t blocks = units * tp.row_tiles;
    if (blocks > 0x7fffffff) return sgv_fail(SGV_ERR_TOO_LARGE, "upfirdn2d: too many workgroups");
    static const int xcd_on = []() { const char* e = getenv("SGV_TILE_XCD"); return e ? atoi(e) : 1; }();    // SGV_TILE_XCD=0: the row-tile-fastest order of rounds 3-5
    tp.xcd_blocks = xcd_on ? (int)((units / 8) * 8 * tp.row_tiles) : 0;
    tp.lds_amax_word = tile_lds_floats(lpr_log2, cpl);
    const size_t lds = (size_t)(tile_lds_floats(lpr_log2, cpl) + 4) * sizeof(float);
    const dim3 grid((unsigned)blocks);
    const bool f44 = p->f_w == 4 && p->f_h == 4 && p->f_sw == 1 && p->f_sh == 4;
    if (!f44) tp.nt_store = 0;      // the general-filter form has one (plain-store) instantiation
    if (dtype == SGV_F32) launch_tile_t<float>(tp, xtra, epi, 4, f44, grid, lds, stream);
    else if (dtype == SGV_F16) launch_tile_t<sgv_half_t>(tp, xtra, epi, cpl, f44, grid, lds, stream);
    else launch_tile_t<sgv_bf16_t>(tp, xtra, epi, cpl, f44, grid, lds, stream);
    sgv_note_variant(epi == 0 ? SGV_V_ufd_tile : epi == 1 ? SGV_V_ufd_tile_fused1 : epi == 2 ? SGV_V_ufd_tile_fused2 : SGV_V_ufd_tile_fused3);
    return sgv_check_launch("upfirdn2d_tile_kernel");
}

// ---- upfirdn2d_down2_tile_kernel: planning and launch ----
bool down2_geometry(const sgv_upfirdn2d_params* p, int dtype, int* lpr_log2) {
    static const int on = []() { const char* e = getenv("SGV_UFD_TILE2X"); return e ? atoi(e) : 1; }();   // SGV_UFD_TILE2X=0: the lane-exchange kernels of rounds 1-5
    if (!on || dtype == SGV_F64) return false;
    if (p->up_x != 1 || p->up_y != 1 || p->down_x != 2 || p->down_y != 2 || p->f_w > 4 || p->f_h > 4) return false;
    if (p->pad_x0 < 0 || p->pad_x0 > 3 || p->pad_y0 < 0 || p->pad_y0 > 3 || p->in_w < 4) return false;
    if (p->out_w < 8 || p->out_w > 128) return false;
    if (!dense_nchw(p->in_w, p->in_h, p->in_c, p->in_sw, p->in_sh, p->in_sc, p->in_sn)) return false;
    if (!dense_nchw(p->out_w, p->out_h, p->in_c, p->out_sw, p->out_sh, p->out_sc, p->out_sn)) return false;
    int l = 1;
    while ((4 << l) < p->out_w) l++;
    *lpr_log2 = l;
    return true;
}

template <typename T>
void launch_down2_t(const down2_params& dp, bool nt, bool f44, dim3 grid, size_t lds, hipStream_t stream) {
    if (!f44) hipLaunchKernelGGL((upfirdn2d_down2_tile_kernel<T, false, false>), grid, dim3(256), lds, stream, dp);
    else if (nt) hipLaunchKernelGGL((upfirdn2d_down2_tile_kernel<T, true, true>), grid, dim3(256), lds, stream, dp);
    else hipLaunchKernelGGL((upfirdn2d_down2_tile_kernel<T, false, true>), grid, dim3(256), lds, stream, dp);
}

int launch_down2(const sgv_upfirdn2d_params* p, int dtype, int lpr_log2, hipStream_t stream, sgv_launch_scope& scope) {
    down2_params dp{};
    dp.y_amax = dtype == SGV_F32 ? scope.take_amax_sink() : nullptr;
    dp.x = p->x; dp.f = p->f; dp.y = p->y; dp.flip = p->flip; dp.gain = p->gain;
    dp.in_w = p->in_w; dp.in_h = p->in_h; dp.out_w = p->out_w; dp.out_h = p->out_h; dp.planes = p->in_c * p->in_n;
    dp.f_w = p->f_w; dp.f_h = p->f_h; dp.f_sw = p->f_sw; dp.f_sh = p->f_sh;
    dp.pad_x = p->pad_x0; dp.pad_y = p->pad_y0;
    dp.lpr_log2 = lpr_log2;
    dp.row_tiles = (p->out_h + D2_ROWS - 1) / D2_ROWS;
    const int ppw = 64 >> lpr_log2;
    const int64_t units = (dp.planes + ppw - 1) / ppw;
    const int64_t blocks = units * dp.row_tiles;
    if (blocks > 0x7fffffff) return sgv_fail(SGV_ERR_TOO_LARGE, "upfirdn2d: too many workgroups");
    static const int xcd_on = []() { const char* e = getenv("SGV_TILE_XCD"); return e ? atoi(e) : 1; }();
    dp.xcd_blocks = xcd_on ? (int)((units / 8) * 8 * dp.row_tiles) : 0;
    const double out_bytes = (double)p->out_w * p->out_h * dp.planes * sgv_dtype_size(dtype);
    const bool nt = out_bytes > 300e6;
    const bool f44 = p->f_w == 4 && p->f_h == 4 && p->f_sw == 1 && p->f_sh == 4;
    dp.lds_amax_word = down2_lds_floats(lpr_log2);
    const size_t lds = (size_t)(down2_lds_floats(lpr_log2) + 4) * sizeof(float);
    const dim3 grid((unsigned)blocks);
    if (dtype == SGV_F32) launch_down2_t<float>(dp, nt, f44, grid, lds, stream);
    else if (dtype == SGV_F16) launch_down2_t<sgv_half_t>(dp, nt, f44, grid, lds, stream);
    else launch_down2_t<sgv_bf16_t>(dp, nt, f44, grid, lds, stream);
    sgv_note_variant(SGV_V_ufd_tile_down2);
    return sgv_check_launch("upfirdn2d_down2_tile_kernel");
}

// ---- upfirdn2d_up2_tile_kernel: planning and launch ----
bool up2_geometry(const sgv_upfirdn2d_params* p, int dtype, int* lpr_log2) {
    static const int on = []() { const char* e = getenv("SGV_UFD_TILE2X"); return e ? atoi(e) : 1; }();
    if (!on || dtype == SGV_F64) return false;
    if (p->up_x != 2 || p->up_y != 2 || p->down_x != 1 || p->down_y != 1 || p->f_w > 4 || p->f_h > 4) return false;
    if (p->pad_x0 < 0 || p->pad_x0 > 3 || p->pad_y0 < 0 || p->pad_y0 > 3 || p->in_w < 4) return false;
    if (p->out_w < 9 || p->out_w > 256) return false;     // (4 lanes per plane row at least: the lanes of a row segment also write its 2 .. 4 halo words)
    if (!dense_nchw(p->in_w, p->in_h, p->in_c, p->in_sw, p->in_sh, p->in_sc, p->in_sn)) return false;
    if (!dense_nchw(p->out_w, p->out_h, p->in_c, p->out_sw, p->out_sh, p->out_sc, p->out_sn)) return false;
    int l = 2;
    while ((4 << l) < p->out_w) l++;
    *lpr_log2 = l;
    return true;
}

template <typename T>
void launch_up2_t(const up2_params& up, bool nt, dim3 grid, size_t lds, hipStream_t stream) {
    if (up.addend) {
        if (nt) hipLaunchKernelGGL((upfirdn2d_up2_tile_kernel<T, true, true>), grid, dim3(256), lds, stream, up);
        else hipLaunchKernelGGL((upfirdn2d_up2_tile_kernel<T, false, true>), grid, dim3(256), lds, stream, up);
    } else {
        if (nt) hipLaunchKernelGGL((upfirdn2d_up2_tile_kernel<T, true, false>), grid, dim3(256), lds, stream, up);
        else hipLaunchKernelGGL((upfirdn2d_up2_tile_kernel<T, false, false>), grid, dim3(256), lds, stream, up);
    }
}

int launch_up2(const sgv_upfirdn2d_params* p, const void* addend, int dtype, int lpr_log2, hipStream_t stream, sgv_launch_scope& scope) {
    up2_params up{};
    up.y_amax = dtype == SGV_F32 ? scope.take_amax_sink() : nullptr;
    up.x = p->x; up.f = p->f; up.y = p->y; up.addend = addend; up.flip = p->flip; up.gain = p->gain;
    up.in_w = p->in_w; up.in_h = p->in_h; up.out_w = p->out_w; up.out_h = p->out_h; up.planes = p->in_c * p->in_n;
    up.f_w = p->f_w; up.f_h = p->f_h; up.f_sw = p->f_sw; up.f_sh = p->f_sh;
    up.pad_x = p->pad_x0; up.pad_y = p->pad_y0;
    up.lpr_log2 = lpr_log2;
    up.row_tiles = (p->out_h + U2_ROWS - 1) / U2_ROWS;
    const int ppw = 64 >> lpr_log2;
    const int64_t units = (up.planes + ppw - 1) / ppw;
    const int64_t blocks = units * up.row_tiles;
    if (blocks > 0x7fffffff) return sgv_fail(SGV_ERR_TOO_LARGE, "upfirdn2d: too many workgroups");
    static const int xcd_on = []() { const char* e = getenv("SGV_TILE_XCD"); return e ? atoi(e) : 1; }();
    up.xcd_blocks = xcd_on ? (int)((units / 8) * 8 * up.row_tiles) : 0;
    const double out_bytes = (double)p->out_w * p->out_h * up.planes * sgv_dtype_size(dtype);
    const bool nt = out_bytes > 300e6;
    up.lds_amax_word = up2_lds_floats(lpr_log2);
    const size_t lds = (size_t)(up2_lds_floats(lpr_log2) + 4) * sizeof(float);
    const dim3 grid((unsigned)blocks);
    if (dtype == SGV_F32) launch_up2_t<float>(up, nt, grid, lds, stream);
    else if (dtype == SGV_F16) launch_up2_t<sgv_half_t>(up, nt, grid, lds, stream);
    else launch_up2_t<sgv_bf16_t>(up, nt, grid, lds, stream);
    sgv_note_variant(addend ? SGV_V_ufd_tile_up2_add : SGV_V_ufd_tile_up2);
    return sgv_check_launch("upfirdn2d_up2_tile_kernel");
}

int validate(const sgv_upfirdn2d_params* p, int dtype) {

    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d: params is NULL");
    if (sgv_dtype_size(dtype) == 0) return sgv_fail(SGV_ERR_UNSUPPORTED, "upfirdn2d: unknown dtype %d", dtype);
    if (!p->x || !p->f || !p->y) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d: x, f and y must be non-NULL");
    if (p->f_w < 1 || p->f_h < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d: f must be at least 1x1");
    if (p->up_x < 1 || p->up_y < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d: upsampling factor must be at least 1");
    if (p->down_x < 1 || p->down_y < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d: downsampling factor must be at least 1");
    if (p->in_w < 1 || p->in_h < 1 || p->in_c < 1 || p->in_n < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d: empty input");
    if ((int64_t)p->in_w * p->in_h * p->in_c * p->in_n > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "upfirdn2d: x is too large");
    const int ow = (p->in_w * p->up_x + p->pad_x0 + p->pad_x1 - p->f_w + p->down_x) / p->down_x;
    const int oh = (p->in_h * p->up_y + p->pad_y0 + p->pad_y1 - p->f_h + p->down_y) / p->down_y;
    if (ow < 1 || oh < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d: output must be at least 1x1");
    if (ow != p->out_w || oh != p->out_h)
        return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d: out size %dx%d does not match the derived %dx%d", p->out_w, p->out_h, ow, oh);
    if ((int64_t)ow * oh * p->in_c * p->in_n > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "upfirdn2d: output is too large");
    return SGV_OK;
}

template <typename T>
void launch_generic(const generic_params& gp, hipStream_t stream) {
    int64_t blocks = (gp.total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(upfirdn2d_generic_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, gp);
}

}  // namespace

extern "C" int sgv_upfirdn2d_kernel_kind(const sgv_upfirdn2d_params* p, int dtype) {
    int rc = validate(p, dtype);
    if (rc != SGV_OK) return rc;
    int lpr_log2, col_groups, xtra;
    if (tile_geometry(p, dtype, &lpr_log2, &col_groups, &xtra)) return 3;
    if (down2_geometry(p, dtype, &lpr_log2)) return 4;
    if (up2_geometry(p, dtype, &lpr_log2)) return 5;
    lanes_plan lplan;
    if (plan_lanes(p, dtype, &lplan)) return 2;
    rows_plan plan;
    return plan_rows(p, dtype, &plan) ? 1 : 0;
}

extern "C" int sgv_upfirdn2d(const sgv_upfirdn2d_params* p, int dtype, void* stream_) {
    int rc = validate(p, dtype);
    if (rc != SGV_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const double bytes = ((double)p->in_w * p->in_h + (double)p->out_w * p->out_h) * p->in_c * p->in_n * sgv_dtype_size(dtype);

    {
        int lpr_log2, col_groups, xtra, cpl;
        if (tile_geometry(p, dtype, &lpr_log2, &col_groups, &xtra, &cpl)) {   // every up = down = 1 FIR pass: the LDS-tile kernel
            sgv_launch_scope scope(SGV_K_UPFIRDN2D_LANES, stream, bytes);
            return launch_tile(p, nullptr, dtype, lpr_log2, col_groups, xtra, cpl, stream, scope);
        }
    }
    {
        int lpr_log2;
        if (down2_geometry(p, dtype, &lpr_log2)) {   // 2x down-sampling with whole planes in a workgroup's lanes: the down2 tile kernel
            sgv_launch_scope scope(SGV_K_UPFIRDN2D_LANES, stream, bytes);
            return launch_down2(p, dtype, lpr_log2, stream, scope);
        }
        if (up2_geometry(p, dtype, &lpr_log2)) {
            sgv_launch_scope scope(SGV_K_UPFIRDN2D_LANES, stream, bytes);
            return launch_up2(p, nullptr, dtype, lpr_log2, stream, scope);
        }
    }
    lanes_plan lplan;
    if (plan_lanes(p, dtype, &lplan)) {
        sgv_launch_scope scope(SGV_K_UPFIRDN2D_LANES, stream, bytes);
        lplan.lp.y_amax = dtype == SGV_F32 ? scope.take_amax_sink() : nullptr;
        hipLaunchKernelGGL(lplan.fn, dim3((unsigned)lplan.blocks), dim3((unsigned)lplan.threads), 0, stream, lplan.lp);
        sgv_note_variant(lplan.lp.lpr_log2 < 6 ? SGV_V_ufd_lanes_seg : SGV_V_ufd_lanes);
        return sgv_check_launch("upfirdn2d_lanes_kernel");
    }
    rows_plan plan;
    if (plan_rows(p, dtype, &plan)) {
        sgv_launch_scope scope(SGV_K_UPFIRDN2D_ROWS, stream, bytes);
        hipLaunchKernelGGL(plan.fn, dim3((unsigned)plan.blocks), dim3(256), 0, stream, plan.rp);
        sgv_note_variant(SGV_V_ufd_rows);
        return sgv_check_launch("upfirdn2d_rows_kernel");
    }

    generic_params gp;
    gp.x = p->x; gp.f = p->f; gp.y = p->y;
    gp.up_x = p->up_x; gp.up_y = p->up_y; gp.down_x = p->down_x; gp.down_y = p->down_y;
    gp.pad_x0 = p->pad_x0; gp.pad_y0 = p->pad_y0; gp.flip = p->flip; gp.gain = p->gain;
    gp.in_w = p->in_w; gp.in_h = p->in_h; gp.in_c = p->in_c; gp.in_n = p->in_n;
    gp.in_sw = p->in_sw; gp.in_sh = p->in_sh; gp.in_sc = p->in_sc; gp.in_sn = p->in_sn;
    gp.f_w = p->f_w; gp.f_h = p->f_h; gp.f_sw = p->f_sw; gp.f_sh = p->f_sh;
    gp.out_w = p->out_w; gp.out_h = p->out_h;
    gp.out_sw = p->out_sw; gp.out_sh = p->out_sh; gp.out_sc = p->out_sc; gp.out_sn = p->out_sn;
    gp.total = (int64_t)p->out_w * p->out_h * p->in_c * p->in_n;
    gp.chan_minor = (p->in_sc == 1 && p->in_c > 1) ? 1 : 0;
    sgv_launch_scope scope(SGV_K_UPFIRDN2D_GENERIC, stream, bytes);
    switch (dtype) {
        case SGV_F32: launch_generic<float>(gp, stream); break;
        case SGV_F16: launch_generic<sgv_half_t>(gp, stream); break;
        case SGV_BF16: launch_generic<sgv_bf16_t>(gp, stream); break;
        case SGV_F64: launch_generic<double>(gp, stream); break;
        default: return sgv_fail(SGV_ERR_UNSUPPORTED, "upfirdn2d: unknown dtype %d", dtype);
    }
    sgv_note_variant(SGV_V_ufd_generic);
    return sgv_check_launch("upfirdn2d_generic_kernel");
}

extern "C" int sgv_upfirdn2d_fused(const sgv_upfirdn2d_params* p, const sgv_fir_epilogue* e, int dtype, void* stream_) {
    int rc = validate(p, dtype);
    if (rc != SGV_OK) return rc;
    if (!e) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d_fused: epilogue is NULL");
    if (e->mode < 1 || e->mode > 4) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d_fused: mode must be 1 (forward epilogue), 2 (backward prologue), 3 (backward epilogue) or 4 (add)");
    if (e->mode != 4 && e->act != 1 && e->act != 3) return sgv_fail(SGV_ERR_UNSUPPORTED, "upfirdn2d_fused: only linear (1) and lrelu (3) are fusable");
    if (e->mode != 4 && e->act == 3 && e->alpha == 0.f) return sgv_fail(SGV_ERR_UNSUPPORTED, "upfirdn2d_fused: lrelu with alpha 0 is not invertible");
    if (e->mode == 4 && !e->yref) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d_fused: mode 4 needs yref (the other summand)");
    if (e->mode == 2 && (!e->yref || !e->sum_g || !e->sum_gv)) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d_fused: mode 2 needs yref, sum_g and sum_gv");
    if (e->mode == 3 && (!e->yref || !e->sum_g)) return sgv_fail(SGV_ERR_INVALID_ARG, "upfirdn2d_fused: mode 3 needs yref and sum_g");
    hipStream_t stream = (hipStream_t)stream_;
    const double es0 = (double)sgv_dtype_size(dtype);
    const double nin0 = (double)p->in_w * p->in_h * p->in_c * p->in_n, nout0 = (double)p->out_w * p->out_h * p->in_c * p->in_n;
    static const int tile2_on = []() { const char* e = getenv("SGV_UFD_TILE_EPI2"); return e ? atoi(e) : 1; }();   // SGV_UFD_TILE_EPI2=0: mode 2 on the lane-exchange kernel (rounds 2-5)
    if (e->mode == 1 || e->mode == 3 || (e->mode == 2 && tile2_on && p->pad_x0 == 2 && p->pad_y0 == 2 && p->out_w == p->in_w + 1 && p->out_h == p->in_h + 1)) {
        int lpr_log2, col_groups, xtra, cpl;
        if (tile_geometry(p, dtype, &lpr_log2, &col_groups, &xtra, &cpl)) {
            sgv_launch_scope scope(SGV_K_UPFIRDN2D_LANES, stream, (e->mode == 2 ? 2.0 * nin0 : nin0) * es0 + (e->mode == 3 ? 2.0 * nout0 : nout0) * es0);
            return launch_tile(p, e, dtype, lpr_log2, col_groups, xtra, cpl, stream, scope);
        }
    }
    if (e->mode == 4) {
        int lpr_log2;
        if (up2_geometry(p, dtype, &lpr_log2)) {
            sgv_launch_scope scope(SGV_K_UPFIRDN2D_LANES, stream, nin0 * es0 + 2.0 * nout0 * es0);
            return launch_up2(p, e->yref, dtype, lpr_log2, stream, scope);
        }
    }
    lanes_plan lplan;
    if (!plan_lanes(p, dtype, &lplan, e))
        return sgv_fail(SGV_ERR_UNSUPPORTED, "upfirdn2d_fused: geometry/layout not covered by the fused kernel (modes 1, 3: up=down=1 pad 1, mode 2: up=down=1 pad 2, mode 4: up=2 pad 2, 4x4 filter, dense NCHW)");
    const double es = (double)sgv_dtype_size(dtype);
    const double nin = (double)p->in_w * p->in_h * p->in_c * p->in_n, nout = (double)p->out_w * p->out_h * p->in_c * p->in_n;
    const double bytes = (e->mode == 2 ? 2.0 * nin : nin) * es + (e->mode >= 3 ? 2.0 * nout : nout) * es;
    sgv_launch_scope scope(SGV_K_UPFIRDN2D_LANES, stream, bytes);
    lplan.lp.y_amax = (dtype == SGV_F32 && e->mode != 1) ? scope.take_amax_sink() : nullptr;
    hipLaunchKernelGGL(lplan.fn, dim3((unsigned)lplan.blocks), dim3((unsigned)lplan.threads), 0, stream, lplan.lp);
    sgv_note_variant(SGV_V_ufd_lanes_fused1 + (e->mode - 1));
    return sgv_check_launch("upfirdn2d_lanes_kernel (fused)");
}
