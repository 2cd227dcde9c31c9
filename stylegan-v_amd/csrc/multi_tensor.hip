// Multi-tensor element-wise passes of the training iteration: one launch over a LIST of tensors instead of one launch per parameter.
//
// sgv_multi_scale_f32 is the equalised learning-rate scaling of a module's convolution weights and biases,
//     w = self.weight * (self.weight_gain * self.lr_multiplier)        b = self.bias * self.lr_multiplier
// (src/training/layers.py:184-185, once per layer, forward pass and gradient: ~250 launches per iteration at FFS-256), out of place, each tensor with
// its own factor.
//
// sgv_multi_nan_to_num_f32 replaces the per-parameter loop of the reference's training loop,
//     for param in phase.module.parameters(): misc.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5, out=param.grad)
// (src/training/training_loop.py:384-386; ~150 launches per phase at FFS-256).  The tensor table travels in the kernel arguments (no device
// memory, no copy: capture-safe); a workgroup finds its tensor by a binary search over the per-tensor workgroup offsets, which live in SGPRs.
#include "sgv_common.h"

namespace {

constexpr int MT_MAX = 96;          // tensors per launch: 96 * (8 + 8 + 4) bytes of kernel arguments
constexpr int MT_CHUNK = 4096;      // elements per workgroup (256 lanes x 4 float4)

struct mt_table {
    float* ptr[MT_MAX];
    int64_t numel[MT_MAX];
    int32_t first_block[MT_MAX + 1];   // workgroup offsets: tensor t owns workgroups [first_block[t], first_block[t + 1])
    int32_t count;
    float nan, posinf, neginf;
};

__device__ __forceinline__ float fix(float v, float nan, float posinf, float neginf) {
    if (v != v) return nan;
    if (v == __builtin_inff()) return posinf;
    if (v == -__builtin_inff()) return neginf;
    return v;
}

__global__ __launch_bounds__(256) void multi_nan_to_num_kernel(const mt_table tb) {
    const int b = blockIdx.x;
    int lo = 0, hi = tb.count;          // wave-uniform: the table is read with scalar loads
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tb.first_block[mid] <= b) lo = mid; else hi = mid;
    }
    float* p = tb.ptr[lo];
    const int64_t n = tb.numel[lo];
    const int64_t base = (int64_t)(b - tb.first_block[lo]) * MT_CHUNK;
    const int64_t left = n - base;
    if (left >= MT_CHUNK && (((uintptr_t)p) & 15) == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4* q = (f4*)(p + base);
        f4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = q[threadIdx.x + 256 * i];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            bool dirty = false;
            f4 r;
#pragma unroll
            for (int j = 0; j < 4; j++) { r[j] = fix(v[i][j], tb.nan, tb.posinf, tb.neginf); dirty |= !(r[j] == v[i][j]); }
            if (dirty) q[threadIdx.x + 256 * i] = r;     // gradients are finite almost always: the pass is then read-only
        }
    } else {
        for (int64_t i = threadIdx.x; i < left && i < MT_CHUNK; i += 256) {
            const float v = p[base + i], r = fix(v, tb.nan, tb.posinf, tb.neginf);
            if (!(r == v)) p[base + i] = r;
        }
    }
}

constexpr int MS_MAX = 64;          // tensors per launch: 64 * (8 + 8 + 8 + 4 + 4) bytes of kernel arguments

struct ms_table {
    const float* src[MS_MAX];
    float* dst[MS_MAX];
    int64_t numel[MS_MAX];
    float scale[MS_MAX];
    int32_t first_block[MS_MAX + 1];
    int32_t count;
};

__global__ __launch_bounds__(256) void multi_scale_kernel(const ms_table tb) {
    const int b = blockIdx.x;
    int lo = 0, hi = tb.count;          // wave-uniform binary search over the per-tensor workgroup offsets (scalar loads of the argument table)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tb.first_block[mid] <= b) lo = mid; else hi = mid;
    }
    const float* p = tb.src[lo];
    float* q = tb.dst[lo];
    const float sc = tb.scale[lo];
    const int64_t n = tb.numel[lo];
    const int64_t base = (int64_t)(b - tb.first_block[lo]) * MT_CHUNK;
    const int64_t left = n - base;
    if (left >= MT_CHUNK && ((((uintptr_t)p) | ((uintptr_t)q)) & 15) == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* pv = (const f4*)(p + base);
        f4* qv = (f4*)(q + base);
        f4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = pv[threadIdx.x + 256 * i];
#pragma unroll
        for (int i = 0; i < 4; i++) qv[threadIdx.x + 256 * i] = v[i] * sc;     // one rounding per element, as torch's `tensor * python_float`
    } else {
        for (int64_t i = threadIdx.x; i < left && i < MT_CHUNK; i += 256) q[base + i] = p[base + i] * sc;
    }
}

}  // namespace

extern "C" int sgv_multi_nan_to_num_f32(float* const* tensors, const int64_t* numels, int32_t count, float nan, float posinf, float neginf, void* stream_) {
    if (count < 0 || (count > 0 && (!tensors || !numels))) return sgv_fail(SGV_ERR_INVALID_ARG, "multi_nan_to_num: bad tensor list");
    hipStream_t stream = (hipStream_t)stream_;
    int done = 0;
    while (done < count) {
        mt_table tb;
        tb.nan = nan; tb.posinf = posinf; tb.neginf = neginf;
        int k = 0;
        int64_t blocks = 0, bytes = 0;
        while (done < count && k < MT_MAX) {
            const int64_t n = numels[done];
            if (n < 0 || (n > 0 && !tensors[done])) return sgv_fail(SGV_ERR_INVALID_ARG, "multi_nan_to_num: tensor %d is NULL or has a negative size", done);
            const int64_t nb = (n + MT_CHUNK - 1) / MT_CHUNK;
            if (blocks + nb > 0x7fffffff) break;
            if (n > 0) { tb.ptr[k] = tensors[done]; tb.numel[k] = n; tb.first_block[k] = (int32_t)blocks; blocks += nb; bytes += n * 4; k++; }
            done++;
        }
        if (k == 0) { if (done < count) return sgv_fail(SGV_ERR_TOO_LARGE, "multi_nan_to_num: tensor %d is too large", done); break; }
        tb.first_block[k] = (int32_t)blocks;
        tb.count = k;
        sgv_launch_scope scope(SGV_K_MODULATE, stream, (double)bytes);
        hipLaunchKernelGGL(multi_nan_to_num_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tb);
        int rc = sgv_check_launch("multi_nan_to_num_kernel");
        if (rc != SGV_OK) return rc;
    }
    return SGV_OK;
}

extern "C" int sgv_multi_scale_f32(const float* const* src, float* const* dst, const int64_t* numels, const float* scales, int32_t count, void* stream_) {
    if (count < 0 || (count > 0 && (!src || !dst || !numels || !scales))) return sgv_fail(SGV_ERR_INVALID_ARG, "multi_scale: bad tensor list");
    hipStream_t stream = (hipStream_t)stream_;
    int done = 0;
    while (done < count) {
        ms_table tb;
        int k = 0;
        int64_t blocks = 0, bytes = 0;
        while (done < count && k < MS_MAX) {
            const int64_t n = numels[done];
            if (n < 0 || (n > 0 && (!src[done] || !dst[done]))) return sgv_fail(SGV_ERR_INVALID_ARG, "multi_scale: tensor %d is NULL or has a negative size", done);
            const int64_t nb = (n + MT_CHUNK - 1) / MT_CHUNK;
            if (blocks + nb > 0x7fffffff) break;
            if (n > 0) { tb.src[k] = src[done]; tb.dst[k] = dst[done]; tb.numel[k] = n; tb.scale[k] = scales[done]; tb.first_block[k] = (int32_t)blocks; blocks += nb; bytes += n * 8; k++; }
            done++;
        }
        if (k == 0) { if (done < count) return sgv_fail(SGV_ERR_TOO_LARGE, "multi_scale: tensor %d is too large", done); break; }
        tb.first_block[k] = (int32_t)blocks;
        tb.count = k;
        sgv_launch_scope scope(SGV_K_MODULATE, stream, (double)bytes);
        hipLaunchKernelGGL(multi_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tb);
        int rc = sgv_check_launch("multi_scale_kernel");
        if (rc != SGV_OK) return rc;
    }
    return SGV_OK;
}
