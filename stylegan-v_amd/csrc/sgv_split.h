// Operand formats of the matrix-pipe convolution / GEMM kernels (TERMS template parameter of every member of the 3x3 family and of the tiled GEMM):
//
//   TERMS = 1   one bf16 operand per value                              (bf16 tensors; "bf16 products" on fp32 tensors); fp16 tensors: one fp16 operand (format 2 below)
//   TERMS = 3   2-way bf16 split  v = hi + lo, 3 MFMAs per product      (16 mantissa bits per operand: 4.4e-6 of the result's scale, NOT fp32-grade)
//   TERMS = 4   2-way fp16 split of the BLOCK-SCALED value, 3 MFMAs     (22 mantissa bits per operand: fp32-grade -- the default since round 4)
//
// TERMS = 4.  The reference's config-3 arithmetic is strict fp32 (src/training/training_loop.py:129,141-142: allow_tf32 = False).  gfx950 has no
// fp32-rate matrix path (v_mfma_f32_32x32x2_f32 = 1/16 of the 16-bit rate), and a bf16 split needs three terms = six MFMAs to reach 24 bits.  fp16
// carries 11 significant bits, so TWO terms hold 22 of fp32's 24 -- but only inside fp16's narrow exponent range.  So every tensor is multiplied by a
// power of two S = 2^(14 - e), e = floor(log2(B)) for an upper bound B >= max|v| of the tensor (`*_amax` pointers of the C ABI; sgv_absmax computes
// one, a producing kernel can leave one behind), which puts max|v * S| into [2^14, 2^15) < 65504:
//       hi = fp16_rne(v * S),  lo = fp16_rne(v * S - hi),     a * b ~ (a_hi * b_hi + a_hi * b_lo + a_lo * b_hi) / (S_a * S_b)
// * |v * S - hi - lo| <= max(2^-22 |v * S|, 2^-25): elements within 2^-18 of the bound keep 22 bits, smaller ones an absolute error below 2^-39 of
//   the bound (fp16 subnormals; were the matrix pipe to flush them the floor would be 2^-29 of the bound -- still below fp32's own rounding of the sum).
// * the products are exact in the MFMA, the accumulation is the same fp32 accumulation every member of the family has; the dropped lo * lo term is
//   2^-22 of a product.  Scaling by powers of two is exact, and the result is scaled back with v_ldexp_f32 (no overflow of a combined factor).
// * measured against float64 (tests/test_conv3x3_gpu.py, profiles/r04_*): rel. error ~1e-7 of the result's scale where the vendor library's fp32
//   convolutions sit at 1.4-3.5e-7 and the bf16 split at 4.4e-6.  Exact on integer data up to 2^11 (hi alone holds it).
// * a bound that is loose by a factor L only raises the absolute floor to 2^-39 * L of the true maximum: bounds may be products of bounds
//   (x * styles: amax(x) * amax(styles), `x_amax2`).  A bound that is too SMALL overflows fp16 (inf / NaN results): the contract of the pointer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgv_conv {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned sp_u32x4 __attribute__((ext_vector_type(4)));

// exponent e of an upper bound B: 2^e <= B < 2^(e+1); clamped so that every derived scale is a normal float.  B = 0 / subnormal -> -100, inf / NaN -> 128.
__device__ __forceinline__ int amax_exponent(float bound) {
    const int e = (int)((__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu) - 127;
    return e < -100 ? -100 : e;
}
// exponent of the product of two bounds (B1 * B2 < 2^(e1 + e2 + 2)), same clamp
__device__ __forceinline__ int amax_exponent2(float b1, float b2) {
    const int e = amax_exponent(b1) + amax_exponent(b2) + 1;
    return e < -100 ? -100 : (e > 128 ? 128 : e);
}
__device__ __forceinline__ float pow2_float(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }   // -126 <= e <= 127
// the scale that takes a tensor bounded by 2^(e+1) into fp16's range: max |v| * S < 2^15
__device__ __forceinline__ float split_scale(int e) { return pow2_float(14 - e); }
// what undoes the two operand scales on the accumulators: acc * 2^unscale_exponent(ea, eb)
__device__ __forceinline__ int unscale_exponent(int ea, int eb) { return ea + eb - 28; }

__device__ __forceinline__ unsigned sp_pack_bf16(float a, float b) {
    sp_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, sp_bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ unsigned sp_pack_f16(float a, float b) {
    sp_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2));       // RNE (fp16 denormals kept)
}

// Operand format of a kernel instantiation: its TERMS, except that 16-bit tensors (TERMS = 1) whose elements ARE fp16 (IO = 2 of sgv_io16.h) are multiplied
// as fp16 operands on v_mfma_f32_32x32x16_f16 -- format 2, one product, no scale, no rounding of the activations at all (their 11 significant bits; the fp32 master
// weights are rounded to fp16 as the reference rounds them, networks.py:50-52 `w.to(x.dtype)`) -- instead of being rounded to bf16's 8 (rounds 1-4: 2e-3 per convolution).
template <int TERMS, int IO>
constexpr int operand_format() { return (TERMS == 1 && IO == 2) ? 2 : TERMS; }

// two neighbouring values -> one dword of the hi plane and one of the lo plane
template <int TERMS>
__device__ __forceinline__ void split2(float a, float b, float S, unsigned& hi, unsigned& lo) {
    if constexpr (TERMS == 2) {
        // a product formed in fp32 on the way in (x * styles) may leave fp16's range: saturate like the reference's clamp would (conv_clamp = 256 bounds x)
        // (NaN-preserving: v_med3_f32 returns the minimum of its operands when one is a NaN -- a NaN activation entered the product as a finite -65504 and the
        // divergence the reference would show was masked; min / max drop the NaN the same way, so the NaN is put back explicitly)
        const float ca = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), cb = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
        hi = sp_pack_f16(a != a ? a : ca, b != b ? b : cb);
        lo = 0u;
    } else if constexpr (TERMS == 4) {
        a *= S; b *= S;
        const unsigned h = sp_pack_f16(a, b);
        const f16x2 hh = __builtin_bit_cast(f16x2, h);
        hi = h;
        lo = sp_pack_f16(a - (float)hh[0], b - (float)hh[1]);
    } else {
        const unsigned h = sp_pack_bf16(a, b);
        hi = h;
        lo = sp_pack_bf16(a - __builtin_bit_cast(float, h << 16), b - __builtin_bit_cast(float, h & 0xffff0000u));
    }
}

// 8 channel values of one pixel -> hi and lo operand words (S: the tensor's split scale, TERMS = 4 only)
template <int TERMS>
__device__ __forceinline__ void split8t(const float* v, float S, sp_u32x4& hi, sp_u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        unsigned h, l;
        split2<TERMS>(v[2 * j], v[2 * j + 1], S, h, l);
        hi[j] = h;
        lo[j] = l;
    }
}

// D = A (32 x 16) * B (16 x 32) + C on the 16-bit matrix pipe: bf16 operands, or fp16 ones for the block-scaled split
template <int TERMS>
__device__ __forceinline__ sp_f32x16 mma16(sp_u32x4 a, sp_u32x4 b, sp_f32x16 c) {
    if constexpr (TERMS == 4 || TERMS == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sp_bf16x8, a), __builtin_bit_cast(sp_bf16x8, b), c, 0, 0, 0);
}

// the split exponent of a kernel's operand from its bound pointer(s), as a wave-uniform value (read once at kernel start, before any pipelined
// inline-asm load is in flight: a compiler-visible load inside those loops drains vmcnt, see wrw_ws_kernel.h)
template <int TERMS>
__device__ __forceinline__ int operand_exponent(const float* amax, const float* amax2 = nullptr) {
    if constexpr (TERMS != 4) return 0;
    else {
        const float b1 = *amax;
        const int e = amax2 ? amax_exponent2(b1, *amax2) : (amax_exponent(b1) > 128 ? 128 : amax_exponent(b1));
        return __builtin_amdgcn_readfirstlane(e);
    }
}

}  // namespace sgv_conv
