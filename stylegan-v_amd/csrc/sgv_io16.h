// Activation element formats of the convolution family: fp32 (IO = 0), bf16 (1), fp16 (2).
//
// The producer waves of the 3x3 kernels load "four consecutive pixels of one channel" per lane with inline-asm loads and counted waits.  A 16-bit
// tensor keeps the SAME number of load instructions -- dwordx4 becomes dwordx2, a halo dword becomes a ushort -- so that every hand-counted
// `s_waitcnt vmcnt(N)` of those kernels holds unchanged; only the width of the register set and the element extraction differ.  16-bit tensors
// are multiplied as single bf16 operands (TERMS = 1: the value IS a bf16, or an fp16 rounded to bf16 on its way into LDS), accumulated in fp32 and
// stored rounded to nearest-even -- the mixed-precision mode of the reference (`num_fp16_res`, networks.py:227,461; train.py:173-174) with bf16
// as the format BASELINE config 4 names.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgv_io {

typedef float io_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned io_u32x2 __attribute__((ext_vector_type(2)));

template <int IO> struct fmt { static constexpr int ES = 2; };      // bytes per element
template <> struct fmt<0> { static constexpr int ES = 4; };

// ---- four consecutive pixels -------------------------------------------------------------------------------------------------------
template <int IO> struct px4 { io_u32x2 v; };
template <> struct px4<0> { io_f32x4 v; };

template <int IO> __device__ __forceinline__ void px4_load(px4<IO>& r, const void* p) {
    if constexpr (IO == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r.v) : "v"(p) : "memory");
}
template <int IO> __device__ __forceinline__ void px4_pin(px4<IO>& r) { asm volatile("" : "+v"(r.v)); }
template <int IO> __device__ __forceinline__ void px4_opaque(px4<IO>& r) { asm volatile("" : "=v"(r.v)); }

__device__ __forceinline__ float half_bits_to_float(unsigned bits16) {
    const uint16_t b = (uint16_t)bits16;
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
// element i (0..3) as fp32
template <int IO> __device__ __forceinline__ float px4_get(const px4<IO>& r, int i) {
    if constexpr (IO == 0) return r.v[i];
    else {
        const unsigned w = r.v[i >> 1];
        if constexpr (IO == 1) return __builtin_bit_cast(float, (i & 1) ? (w & 0xffff0000u) : (w << 16));
        else return half_bits_to_float((i & 1) ? (w >> 16) : (w & 0xffffu));
    }
}

// ---- one pixel ---------------------------------------------------------------------------------------------------------------------
template <int IO> struct px1 { unsigned v; };
template <> struct px1<0> { float v; };

template <int IO> __device__ __forceinline__ void px1_load(px1<IO>& r, const void* p) {
    if constexpr (IO == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(r.v) : "v"(p) : "memory");
    else asm volatile("global_load_ushort %0, %1, off" : "=v"(r.v) : "v"(p) : "memory");
}
template <int IO> __device__ __forceinline__ void px1_pin(px1<IO>& r) { asm volatile("" : "+v"(r.v)); }
template <int IO> __device__ __forceinline__ void px1_opaque(px1<IO>& r) { asm volatile("" : "=v"(r.v)); }
template <int IO> __device__ __forceinline__ float px1_get(const px1<IO>& r) {
    if constexpr (IO == 0) return r.v;
    else if constexpr (IO == 1) return __builtin_bit_cast(float, r.v << 16);
    else return half_bits_to_float(r.v & 0xffffu);
}

// ---- stores ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float v) {   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    typedef float io_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 io_bf16x2 __attribute__((ext_vector_type(2)));
    const io_f32x2 f = {v, 0.f};
    return (uint16_t)__builtin_bit_cast(unsigned, __builtin_convertvector(f, io_bf16x2));
}
template <int IO> __device__ __forceinline__ void out_store(void* base, size_t index, float v) {
    if constexpr (IO == 0) ((float*)base)[index] = v;
    else if constexpr (IO == 1) ((uint16_t*)base)[index] = f32_to_bf16_rne(v);
    else { const _Float16 h = (_Float16)v; uint16_t b; __builtin_memcpy(&b, &h, 2); ((uint16_t*)base)[index] = b; }
}
// the same with a wave-uniform byte base (scalar registers) and a 32-bit unsigned per-lane ELEMENT offset: the scalar-base addressing mode, no 64-bit vector address
template <int IO> __device__ __forceinline__ void out_store_lane(char* ubase, unsigned lane_index, float v) {
    if constexpr (IO == 0) ((float*)ubase)[lane_index] = v;
    else if constexpr (IO == 1) ((uint16_t*)ubase)[lane_index] = f32_to_bf16_rne(v);
    else { const _Float16 h = (_Float16)v; uint16_t b; __builtin_memcpy(&b, &h, 2); ((uint16_t*)ubase)[lane_index] = b; }
}
// element pointer arithmetic on an untyped base
template <int IO> __device__ __forceinline__ const char* at(const void* base, size_t index) { return (const char*)base + index * fmt<IO>::ES; }

}  // namespace sgv_io
