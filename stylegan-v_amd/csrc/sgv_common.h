// Shared helpers for libsgv_hip.so (gfx950 only).  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sgv_ops.h"

// ---------------------------------------------------------------------------------------------
// Storage types.  fp16/bf16 are moved as raw 16-bit words and widened to fp32 in registers
// (fp32 internal accumulate, the reference's InternalType<half>=float: upfirdn2d.cu:15-18).

struct sgv_half_t { uint16_t bits; };
struct sgv_bf16_t { uint16_t bits; };

template <typename T> struct sgv_traits;
template <> struct sgv_traits<float> {
    typedef float acc_t;
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct sgv_traits<double> {
    typedef double acc_t;
    static __device__ __forceinline__ double load(const double* p) { return *p; }
    static __device__ __forceinline__ void store(double* p, double v) { *p = v; }
};
template <> struct sgv_traits<sgv_half_t> {
    typedef float acc_t;
    static __device__ __forceinline__ float load(const sgv_half_t* p) {
        _Float16 h; __builtin_memcpy(&h, p, 2); return (float)h;
    }
    static __device__ __forceinline__ void store(sgv_half_t* p, float v) {
        _Float16 h = (_Float16)v; __builtin_memcpy(p, &h, 2);  // v_cvt_f16_f32: round-to-nearest-even
    }
};
template <> struct sgv_traits<sgv_bf16_t> {
    typedef float acc_t;
    static __device__ __forceinline__ float load(const sgv_bf16_t* p) {
        return __builtin_bit_cast(float, (uint32_t)p->bits << 16);
    }
    static __device__ __forceinline__ void store(sgv_bf16_t* p, float v) {
        uint32_t u = __builtin_bit_cast(uint32_t, v);
        if ((u & 0x7fffffffu) > 0x7f800000u) { p->bits = (uint16_t)((u >> 16) | 0x40); return; }  // quiet NaN
        u += 0x7fffu + ((u >> 16) & 1u);  // round-to-nearest-even
        p->bits = (uint16_t)(u >> 16);
    }
};

static inline size_t sgv_dtype_size(int dtype) {
    switch (dtype) {
        case SGV_F32: return 4;
        case SGV_F16: return 2;
        case SGV_BF16: return 2;
        case SGV_F64: return 8;
        default: return 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Error reporting / launch accounting / profiling (implemented in sgv_runtime.hip).

int sgv_fail(int code, const char* fmt, ...);

struct sgv_launch_scope {
    // Brackets one kernel launch: bumps the launch counter and, when profiling is enabled,
    // records a start/stop HIP event pair on `stream`.
    // `count` = false: a helper pass that is timed but not counted by sgv_launch_count (the tests pin launch counts of the ops themselves).
    // `own_stamps` = true: the launcher's kernel can write the record's timestamp pair ITSELF (first workgroup: start; every workgroup's last wave: atomicMax of
    // the end).  When the launch is being captured with profiling on, the scope then adds no timestamp kernels around it -- no idle gap, nothing for the chip's power
    // management to react to -- and the launcher passes kernel_stamps() to its kernel; a launcher that ends up on a kernel without that support calls begin_stamp()
    // before its launch instead (the ordinary bracket).
    sgv_launch_scope(int family, hipStream_t stream, double bytes, double flops = 0.0, bool count = true, bool own_stamps = false);
    ~sgv_launch_scope();
    unsigned long long* kernel_stamps();     // device pointer to {start, end} of this record for the kernel to fill, or NULL (not capturing / not recording)
    void begin_stamp();                      // own_stamps scopes only: fall back to the bracketing timestamp kernel
    int slot;
    int stamp_mode;     // 0 events / nothing, 1 bracketing timestamp kernels, 2 deferred (own_stamps, undecided), 3 the kernel writes the pair
    hipStream_t stream;
    // The one-shot magnitude-bound side output armed by sgv_amax_sink() for THIS call (moved out of the thread's slot by the constructor, so that a call
    // that cannot serve it leaves it unserved instead of handing it to a later one).  A launcher that supports it calls take_amax_sink(): the pointer
    // (it holds 0.0f -- or a bound to extend -- by the caller's contract: the kernel folds max |output| into it with atomicMax), or NULL when nothing was armed.
    float* amax_sink;
    float* amax_taken;      // the sink a kernel of this call writes its partial maxima to: the destructor folds them into [0]
    float* take_amax_sink();
};

// |v| folded into a running maximum as an fp32 bit pattern.  Protocol of a kernel with a bound side output `sink` (may be NULL):
//   amx = sgv_amax_fold(amx, v);        per stored value (one VALU operation)
//   sgv_amax_commit(amx, sink);         at the end, by EVERY lane of the wave: wave reduction, then ONE no-return atomic per wave into one of
//                                       SGV_AMAX_SLOTS partial slots behind sink[0] (sink[1 + wave id % SLOTS]).
// A streaming kernel with one 16-byte vector per lane has ~10^6 waves per launch: with a single target address -- even behind a "read it first, add only
// if larger" filter -- those requests queue on ONE L2 channel and cost the kernel 40-70 % (measured, profiles/r04 call 2 / 4); spread over 4,096 slots
// (128 cache lines) they disappear in the kernel's own traffic.  The launch scope (sgv_launch_scope) folds the slots into sink[0] with a one-workgroup
// kernel behind the producer.  The sink block (1 + SGV_AMAX_SLOTS floats) must hold zeros when the producer starts.
constexpr int SGV_AMAX_SLOTS = 4096;
__device__ __forceinline__ unsigned sgv_amax_fold(unsigned m, float v) { return max(m, __builtin_bit_cast(unsigned, v) & 0x7fffffffu); }
__device__ __forceinline__ void sgv_amax_commit(unsigned m, float* sink) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m) {
        const unsigned wid = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        atomicMax((unsigned*)sink + 1 + (wid & (SGV_AMAX_SLOTS - 1)), m);
    }
}

// The same with ONE atomic per workgroup: the waves fold their maxima into an LDS word first (ds_max_u32).  `lds_word` holds 0 from before the workgroup's
// last barrier; every thread of the workgroup calls this (it contains a barrier).  Round 6: the tile kernels of upfirdn2d retire a wave per 4 output rows --
// four times the atomics of the strip walkers -- and the armed side output cost the 2x up-sampling pass 11 % (profiles/r06_c11_*).
__device__ __forceinline__ void sgv_amax_commit_wg(unsigned m, float* sink, unsigned* lds_word) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(lds_word, m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned w = *lds_word;
        if (w) atomicMax((unsigned*)sink + 1 + (blockIdx.x & (SGV_AMAX_SLOTS - 1)), w);
    }
}

// Kernel variants (sgv_variant_count / sgv_variant_name of the public header): which member of a family a call took.
#define SGV_VARIANTS(X) \
    X(conv_s1_ws) X(conv_s1_ws_fused) X(conv_s1_ws_accumulate) X(conv_s1_4wave) X(conv_small) \
    X(conv_s2_pairs) X(conv_s2_pairs_fused) X(conv_s2_pairs_packed) X(conv_s2_pairs_packed_fused) X(conv_s2_ws) X(conv_s2_1role) \
    X(convT_ws) X(convT_ws_packed) X(convT_1role) X(convT_edge_mfma) X(convT_edge_gather) \
    X(wrw_s1_ws) X(wrw_s1_ws_scaled) X(wrw_s1_ws_packed) X(wrw_s1_4wave) X(wrw_s2_ws) X(wrw_s2_ws_packed) X(wrw_s2_4wave) X(wrw_s2_4wave_packed) \
    X(ufd_tile) X(ufd_tile_fused1) X(ufd_tile_fused3) X(ufd_lanes) X(ufd_lanes_seg) X(ufd_lanes_fused1) X(ufd_lanes_fused2) X(ufd_lanes_fused3) X(ufd_lanes_fused4) X(ufd_rows) X(ufd_generic) \
    X(pw_many2few) X(pw_few2many) X(pw_few2many_act) X(pw_outer) X(gemm_f32) X(gemm_bf16x3) X(fc) X(bias_act) X(conv_lowp) X(wrw_lowp) X(conv_s2_lowp) X(convT_lowp) X(wrw_s2_lowp) X(gemm_bf16x3_stream) X(conv1x1_wstat) \
    X(conv_s1_half_tile) X(convT_half_tile) X(ufd_tile_down2) X(ufd_tile_up2) X(ufd_tile_up2_add) X(ufd_tile_fused2) X(fc_grouped)
enum sgv_variant_id {
#define SGV_V_ENUM(name) SGV_V_##name,
    SGV_VARIANTS(SGV_V_ENUM)
#undef SGV_V_ENUM
    SGV_V_COUNT
};
void sgv_note_variant(int v);

static inline int sgv_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SGV_OK;
}
