// Weight (de)modulation arithmetic of modulated_conv2d (src/training/networks.py:57-74) for gfx950.
//
// The reference materialises w[N,O,I,kh,kw] = W * s just to reduce it to d[N,O] (906 MB at N=96 for
// a 512->512 3x3 layer).  Algebraically
//     d[n,o] = rsqrt( sum_i s[n,i]^2 * q[o,i] + eps ),   q[o,i] = sum_k W[o,i,k]^2
// so three small kernels replace it:
//   weight_sqsum_kernel  q[o,i]                 HBM-bound over the weights: (kk+1)*O*I*4 bytes
//   demod_coefs_kernel   d[n,o]                 one wave per output, lanes stride over i, DPP/shuffle
//                                               butterfly reduction (the "per-style demodulation
//                                               reduction"); q rows are L2-resident
//   scale_channels_kernel y[n,c,:] = x[n,c,:] * s[n,c]   HBM stream, 16 B per lane
//
// Algorithmic bytes: sqsum (kk+1)*O*I*4; demod (O*I + N*I + N*O)*4; scale 2*numel(x)*sizeof(T).

#include "sgv_common.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(256) void weight_sqsum_kernel(const float* __restrict__ w, float* __restrict__ q, int total, int kk) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float* p = w + (size_t)idx * kk;
    float acc = 0.f;
    for (int k = 0; k < kk; k++) acc = __builtin_fmaf(p[k], p[k], acc);
    q[idx] = acc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// grid = (ceil(O/4), N); block = 256 = 4 waves; wave w of block bx handles o = bx*4 + w for sample n = by.
__global__ __launch_bounds__(256) void demod_coefs_kernel(const float* __restrict__ s, const float* __restrict__ q,
                                                          float* __restrict__ d, int n_samples, int oc, int ic, float eps) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.y;
    if (o >= oc) return;
    const float* sr = s + (size_t)n * ic;
    const float* qr = q + (size_t)o * ic;
    float acc = 0.f;
    for (int i = lane; i < ic; i += 64) {
        float sv = sr[i];
        acc = __builtin_fmaf(sv * sv, qr[i], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) d[(size_t)n * oc + o] = 1.0f / sqrtf(acc + eps);
}

template <typename T> struct vec16 {
    static constexpr int N = 16 / sizeof(T);
    T e[N];
} __attribute__((aligned(16)));

// Vector path: hw % N == 0 so a 16-B vector never straddles two planes.
// y_amax (fp32 tensors, or NULL): max |y| as a by-product for the block-scaled split of the convolution that reads y (sgv_amax_sink)
template <typename T>
__global__ __launch_bounds__(256) void scale_channels_vec_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                                 T* __restrict__ y, int nvec, int hw, float* y_amax) {
    constexpr int N = vec16<T>::N;
    const vec16<T>* xv = (const vec16<T>*)x;
    vec16<T>* yv = (vec16<T>*)y;
    unsigned amx = 0u;
    for (int vi = blockIdx.x * blockDim.x + threadIdx.x; vi < nvec; vi += gridDim.x * blockDim.x) {
        const float sc = s[(vi * N) / hw];
        vec16<T> v = xv[vi], o;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const float r = sgv_traits<T>::load(&v.e[k]) * sc;
            sgv_traits<T>::store(&o.e[k], r);
            if constexpr (sizeof(T) == 4) amx = sgv_amax_fold(amx, r);
        }
        yv[vi] = o;
    }
    if constexpr (sizeof(T) == 4) { if (y_amax) sgv_amax_commit(amx, y_amax); }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_channels_scalar_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                                    T* __restrict__ y, int total, int hw) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        sgv_traits<T>::store(y + i, sgv_traits<T>::load(x + i) * s[i / hw]);
}

template <typename T>
void launch_scale(const void* x, const float* s, void* y, int total, int hw, hipStream_t stream, sgv_launch_scope& scope) {
    constexpr int N = vec16<T>::N;
    const bool vec_ok = (hw % N == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    float* y_amax = (vec_ok && sizeof(T) == 4) ? scope.take_amax_sink() : nullptr;
    int work = vec_ok ? total / N : total;
    int blocks = (work + 255) / 256;  // one vector per lane, grid covers the tensor (see bias_act.hip)
    if (blocks < 1) blocks = 1;
    if (vec_ok)
        hipLaunchKernelGGL(scale_channels_vec_kernel<T>, dim3(blocks), dim3(256), 0, stream, (const T*)x, s, (T*)y, work, hw, y_amax);
    else
        hipLaunchKernelGGL(scale_channels_scalar_kernel<T>, dim3(blocks), dim3(256), 0, stream, (const T*)x, s, (T*)y, total, hw);
}

// out[p] += sum over the plane's pixels of a[p,:] * b[p,:] (fp32): the styles gradient of x * s[n,c] without materialising a * b.
// grid = (chunks per plane, planes); a workgroup reduces CHUNK = 4096 elements (wave shuffle, LDS across the 4 waves) and issues ONE atomic.
constexpr int PD_CHUNK = 4096;

template <typename T>
__global__ __launch_bounds__(256) void plane_dot_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ out, int hw, int vec_ok) {
    constexpr int N = vec16<T>::N;
    const int plane = blockIdx.y;
    const int p0 = blockIdx.x * PD_CHUNK, p1 = min(hw, p0 + PD_CHUNK);
    const T* ap = a + (size_t)plane * hw;
    const T* bp = b + (size_t)plane * hw;
    float acc = 0.f;
    if (vec_ok) {
        for (int i = p0 + threadIdx.x * N; i < p1; i += 256 * N) {
            const vec16<T> va = *(const vec16<T>*)(ap + i), vb = *(const vec16<T>*)(bp + i);
#pragma unroll
            for (int k = 0; k < N; k++) acc = __builtin_fmaf((float)sgv_traits<T>::load(&va.e[k]), (float)sgv_traits<T>::load(&vb.e[k]), acc);
        }
    } else {
        for (int i = p0 + threadIdx.x; i < p1; i += 256) acc = __builtin_fmaf((float)sgv_traits<T>::load(ap + i), (float)sgv_traits<T>::load(bp + i), acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + plane, (part[0] + part[1]) + (part[2] + part[3]));
}

template <typename T>
void launch_plane_dot(const void* a, const void* b, float* out, int planes, int hw, hipStream_t stream) {
    constexpr int N = vec16<T>::N;
    const int vec_ok = (hw % N == 0) && (((uintptr_t)a | (uintptr_t)b) % 16 == 0);
    dim3 grid((unsigned)((hw + PD_CHUNK - 1) / PD_CHUNK), (unsigned)planes);
    hipLaunchKernelGGL(plane_dot_kernel<T>, grid, dim3(256), 0, stream, (const T*)a, (const T*)b, out, hw, vec_ok);
}

// Backward prologue of a fused "conv -> * d[n,c] -> + bias -> lrelu/linear -> * gain -> clamp" layer (csrc/conv3x3_ws_kernel.h, EPI = 1), one pass:
//   dz    = bias_act gradient (bias_act.cu:39-146, grad = 1) of the incoming dy, from the saved OUTPUT y: ((y > 0) ? dy : dy * alpha) * gain, 0 where clamped
//   out   = dz * d[plane]                               gradient w.r.t. the convolution result
//   sums[0][plane] += sum dz                            -> bias gradient (summed over samples by the caller)
//   sums[1][plane] += sum dz * v,  v = pre-activation   -> demodulation-coefficient gradient: (sums[1] - bias * sums[0]) / d
// dz * v needs no inverse activation: dz = dy * gain * slope and v = y / (gain * slope), so dz * v = dy * y wherever dz != 0.
// T: element type of dy / y / out (d and the sums are fp32); 16-bit tensors: arithmetic in fp32, one rounding on the store.
template <typename T>
__global__ __launch_bounds__(256) void act_grad_scale_kernel(const T* __restrict__ dy, const T* __restrict__ y, const float* __restrict__ d,
                                                             T* __restrict__ out, float* __restrict__ sums, int planes, int hw, int act, float alpha, float gain,
                                                             float clamp, int vec_ok, float* out_amax) {
    constexpr int N = vec16<T>::N;
    const int plane = blockIdx.y;
    const int p0 = blockIdx.x * PD_CHUNK, p1 = min(hw, p0 + PD_CHUNK);
    const T* gp = dy + (size_t)plane * hw;
    const T* yp = y + (size_t)plane * hw;
    T* op = out + (size_t)plane * hw;
    const float dsc = d ? d[plane] : 1.f;
    float sg = 0.f, sgv = 0.f;
    unsigned amx = 0u;      // fp32 tensors: max |out| as a by-product (out_amax, sgv_amax_sink)
    auto one = [&](float g, float yy) {
        float dz = ((act == 3 && !(yy > 0.f)) ? g * alpha : g) * gain;
        float gv = g * yy;
        if (clamp >= 0.f && !(yy > -clamp & yy < clamp)) { dz = 0.f; gv = 0.f; }
        sg += dz;
        sgv += gv;
        const float r = dz * dsc;
        if constexpr (sizeof(T) == 4) amx = sgv_amax_fold(amx, r);
        return r;
    };
    if (vec_ok) {
        for (int i = p0 + threadIdx.x * N; i < p1; i += 256 * N) {
            const vec16<T> g = *(const vec16<T>*)(gp + i), yy = *(const vec16<T>*)(yp + i);
            vec16<T> o;
#pragma unroll
            for (int k = 0; k < N; k++) sgv_traits<T>::store(&o.e[k], one(sgv_traits<T>::load(&g.e[k]), sgv_traits<T>::load(&yy.e[k])));
            *(vec16<T>*)(op + i) = o;
        }
    } else {
        for (int i = p0 + threadIdx.x; i < p1; i += 256) sgv_traits<T>::store(op + i, one(sgv_traits<T>::load(gp + i), sgv_traits<T>::load(yp + i)));
    }
    if constexpr (sizeof(T) == 4) { if (out_amax) sgv_amax_commit(amx, out_amax); }
    if (!sums) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { sg += __shfl_xor(sg, off, 64); sgv += __shfl_xor(sgv, off, 64); }
    __shared__ float part[8];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = sg; part[4 + (threadIdx.x >> 6)] = sgv; }
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums + plane, (part[0] + part[1]) + (part[2] + part[3]));
    if (threadIdx.x == 64) atomicAdd(sums + planes + plane, (part[4] + part[5]) + (part[6] + part[7]));
}

// out = a * s[plane] and dot[plane] += sum a * b in one pass: the input gradient dx = dxs * styles of a modulated layer together with the
// styles gradient sum_px dxs * x (networks.py:66), instead of plane_dot + scale_channels (5 tensor passes -> 3).
// ADD: out = a * s + addend (the gradient that arrived from the tensor's other consumer: the sum lands in this pass's store instead of a separate full-tensor addition)
template <typename T, bool ADD>
__global__ __launch_bounds__(256) void scale_dot_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ s, const T* __restrict__ addend, T* __restrict__ out,
                                                        float* __restrict__ dot, int hw, int vec_ok) {
    constexpr int N = vec16<T>::N;
    const int plane = blockIdx.y;
    const int p0 = blockIdx.x * PD_CHUNK, p1 = min(hw, p0 + PD_CHUNK);
    const T* ap = a + (size_t)plane * hw;
    const T* bp = b + (size_t)plane * hw;
    const T* dp = ADD ? addend + (size_t)plane * hw : nullptr;
    T* op = out + (size_t)plane * hw;
    const float sc = s[plane];
    float acc = 0.f;
    if (vec_ok) {
        for (int i = p0 + threadIdx.x * N; i < p1; i += 256 * N) {
            const vec16<T> va = *(const vec16<T>*)(ap + i), vb = *(const vec16<T>*)(bp + i);
            vec16<T> vd, o;
            if constexpr (ADD) vd = *(const vec16<T>*)(dp + i);
#pragma unroll
            for (int k = 0; k < N; k++) {
                const float av = sgv_traits<T>::load(&va.e[k]);
                acc = __builtin_fmaf(av, sgv_traits<T>::load(&vb.e[k]), acc);
                float r = av * sc;
                if constexpr (ADD) r += sgv_traits<T>::load(&vd.e[k]);
                sgv_traits<T>::store(&o.e[k], r);
            }
            *(vec16<T>*)(op + i) = o;
        }
    } else {
        for (int i = p0 + threadIdx.x; i < p1; i += 256) {
            const float va = sgv_traits<T>::load(ap + i);
            acc = __builtin_fmaf(va, sgv_traits<T>::load(bp + i), acc);
            float r = va * sc;
            if constexpr (ADD) r += sgv_traits<T>::load(dp + i);
            sgv_traits<T>::store(op + i, r);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dot + plane, (part[0] + part[1]) + (part[2] + part[3]));
}

}  // namespace

extern "C" int sgv_act_grad_scale_t(const void* dy, const void* y, const float* d, void* out, float* sums, int32_t planes, int32_t hw, int32_t act, float alpha,
                                    float gain, float clamp, int dtype, void* stream_) {
    if (!dy || !y || !out) return sgv_fail(SGV_ERR_INVALID_ARG, "act_grad_scale: NULL pointer");
    if (planes < 1 || hw < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "act_grad_scale: needs planes >= 1, hw >= 1");
    if ((int64_t)planes * hw > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "act_grad_scale: tensors are too large");
    if (act != 1 && act != 3) return sgv_fail(SGV_ERR_INVALID_ARG, "act_grad_scale: act must be 1 (linear) or 3 (lrelu)");
    const size_t es = sgv_dtype_size(dtype);
    if (es == 0 || dtype == SGV_F64) return sgv_fail(SGV_ERR_UNSUPPORTED, "act_grad_scale: unsupported dtype %d", dtype);
    hipStream_t stream = (hipStream_t)stream_;
    sgv_launch_scope scope(SGV_K_MODULATE, stream, 3.0 * planes * (double)hw * (double)es);
    const int vec_ok = (hw % (16 / (int)es) == 0) && (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) % 16 == 0);
    float* out_amax = dtype == SGV_F32 ? scope.take_amax_sink() : nullptr;
    // blockIdx.y is limited to 65535: the planes go in slabs (one launch up to 65,535 planes = 127 frames of a 512-channel layer; more than that -- the
    // Dmain phase as one pass over generated + real clips, larger per-GPU batches -- takes further launches on the following planes)
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = std::min(65535, planes - p0);
        const size_t off = (size_t)p0 * hw * es;
        const char *dyp = (const char*)dy + off, *yp = (const char*)y + off;
        char* op = (char*)out + off;
        const float* dp = d ? d + p0 : nullptr;
        float* sp = sums ? sums + p0 : nullptr;     // (the kernel's second row sits `planes` behind the first: the total, not the slab)
        dim3 grid((unsigned)((hw + PD_CHUNK - 1) / PD_CHUNK), (unsigned)np);
        if (dtype == SGV_F32) hipLaunchKernelGGL(act_grad_scale_kernel<float>, grid, dim3(256), 0, stream, (const float*)dyp, (const float*)yp, dp, (float*)op, sp, planes, hw, act, alpha, gain, clamp, vec_ok, out_amax);
        else if (dtype == SGV_F16) hipLaunchKernelGGL(act_grad_scale_kernel<sgv_half_t>, grid, dim3(256), 0, stream, (const sgv_half_t*)dyp, (const sgv_half_t*)yp, dp, (sgv_half_t*)op, sp, planes, hw, act, alpha, gain, clamp, vec_ok, nullptr);
        else hipLaunchKernelGGL(act_grad_scale_kernel<sgv_bf16_t>, grid, dim3(256), 0, stream, (const sgv_bf16_t*)dyp, (const sgv_bf16_t*)yp, dp, (sgv_bf16_t*)op, sp, planes, hw, act, alpha, gain, clamp, vec_ok, nullptr);
    }
    return sgv_check_launch("act_grad_scale_kernel");
}

extern "C" int sgv_act_grad_scale(const float* dy, const float* y, const float* d, float* out, float* sums, int32_t planes, int32_t hw, int32_t act, float alpha,
                                  float gain, float clamp, void* stream_) {
    return sgv_act_grad_scale_t(dy, y, d, out, sums, planes, hw, act, alpha, gain, clamp, SGV_F32, stream_);
}

template <typename T>
static void launch_scale_dot(const char* ap, const char* bp, const float* s, const char* dp, char* op, float* dot, int hw, int vec_ok, dim3 grid, hipStream_t stream) {
    if (dp) hipLaunchKernelGGL((scale_dot_kernel<T, true>), grid, dim3(256), 0, stream, (const T*)ap, (const T*)bp, s, (const T*)dp, (T*)op, dot, hw, vec_ok);
    else hipLaunchKernelGGL((scale_dot_kernel<T, false>), grid, dim3(256), 0, stream, (const T*)ap, (const T*)bp, s, (const T*)nullptr, (T*)op, dot, hw, vec_ok);
}

extern "C" int sgv_scale_dot_add_t(const void* a, const void* b, const float* s, const void* addend, void* out, float* dot, int32_t planes, int32_t hw, int dtype, void* stream_) {
    if (!a || !b || !s || !out || !dot) return sgv_fail(SGV_ERR_INVALID_ARG, "scale_dot: NULL pointer");
    if (planes < 1 || hw < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "scale_dot: needs planes >= 1, hw >= 1");
    if ((int64_t)planes * hw > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "scale_dot: tensors are too large");
    const size_t es = sgv_dtype_size(dtype);
    if (es == 0 || dtype == SGV_F64) return sgv_fail(SGV_ERR_UNSUPPORTED, "scale_dot: unsupported dtype %d", dtype);
    hipStream_t stream = (hipStream_t)stream_;
    sgv_launch_scope scope(SGV_K_MODULATE, stream, (addend ? 4.0 : 3.0) * planes * (double)hw * (double)es);
    const int vec_ok = (hw % (16 / (int)es) == 0) && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)addend) % 16 == 0);
    for (int p0 = 0; p0 < planes; p0 += 65535) {     // blockIdx.y <= 65535: planes in slabs, as act_grad_scale
        const int np = std::min(65535, planes - p0);
        const size_t off = (size_t)p0 * hw * es;
        const char *ap = (const char*)a + off, *bp = (const char*)b + off, *dp = addend ? (const char*)addend + off : nullptr;
        char* op = (char*)out + off;
        dim3 grid((unsigned)((hw + PD_CHUNK - 1) / PD_CHUNK), (unsigned)np);
        if (dtype == SGV_F32) launch_scale_dot<float>(ap, bp, s + p0, dp, op, dot + p0, hw, vec_ok, grid, stream);
        else if (dtype == SGV_F16) launch_scale_dot<sgv_half_t>(ap, bp, s + p0, dp, op, dot + p0, hw, vec_ok, grid, stream);
        else launch_scale_dot<sgv_bf16_t>(ap, bp, s + p0, dp, op, dot + p0, hw, vec_ok, grid, stream);
    }
    return sgv_check_launch("scale_dot_kernel");
}

extern "C" int sgv_scale_dot_t(const void* a, const void* b, const float* s, void* out, float* dot, int32_t planes, int32_t hw, int dtype, void* stream_) {
    return sgv_scale_dot_add_t(a, b, s, nullptr, out, dot, planes, hw, dtype, stream_);
}

extern "C" int sgv_scale_dot(const float* a, const float* b, const float* s, float* out, float* dot, int32_t planes, int32_t hw, void* stream_) {
    return sgv_scale_dot_t(a, b, s, out, dot, planes, hw, SGV_F32, stream_);
}

extern "C" int sgv_plane_dot(const void* a, const void* b, float* out, int32_t planes, int32_t hw, int dtype, void* stream_) {
    if (!a || !b || !out) return sgv_fail(SGV_ERR_INVALID_ARG, "plane_dot: NULL pointer");
    if (planes < 1 || hw < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "plane_dot: sizes must be positive");
    if (planes > 65535 * 16) return sgv_fail(SGV_ERR_TOO_LARGE, "plane_dot: too many planes");
    if ((int64_t)planes * hw > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "plane_dot: tensors are too large");
    const size_t es = sgv_dtype_size(dtype);
    if (es == 0 || dtype == SGV_F64) return sgv_fail(SGV_ERR_UNSUPPORTED, "plane_dot: unsupported dtype %d", dtype);
    hipStream_t stream = (hipStream_t)stream_;
    sgv_launch_scope scope(SGV_K_MODULATE, stream, 2.0 * planes * (double)hw * es + planes * 4.0);
    // blockIdx.y is limited to 65535: fold the planes in slabs
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = std::min(65535, planes - p0);
        const char* ap = (const char*)a + (size_t)p0 * hw * es;
        const char* bp = (const char*)b + (size_t)p0 * hw * es;
        switch (dtype) {
            case SGV_F32: launch_plane_dot<float>(ap, bp, out + p0, np, hw, stream); break;
            case SGV_F16: launch_plane_dot<sgv_half_t>(ap, bp, out + p0, np, hw, stream); break;
            default: launch_plane_dot<sgv_bf16_t>(ap, bp, out + p0, np, hw, stream); break;
        }
    }
    return sgv_check_launch("plane_dot_kernel");
}

extern "C" int sgv_weight_sqsum(const float* w, float* wsq, int32_t oc, int32_t ic, int32_t kk, void* stream_) {
    if (!w || !wsq) return sgv_fail(SGV_ERR_INVALID_ARG, "weight_sqsum: NULL pointer");
    if (oc < 1 || ic < 1 || kk < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "weight_sqsum: sizes must be positive");
    if ((int64_t)oc * ic * kk > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "weight_sqsum: weight is too large");
    hipStream_t stream = (hipStream_t)stream_;
    const int total = oc * ic;
    sgv_launch_scope scope(SGV_K_MODULATE, stream, (double)total * (kk + 1) * 4.0);
    hipLaunchKernelGGL(weight_sqsum_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, wsq, total, kk);
    return sgv_check_launch("weight_sqsum_kernel");
}

extern "C" int sgv_demod_coefs(const float* styles, const float* wsq, float* dcoefs, int32_t n, int32_t oc, int32_t ic,
                               float eps, void* stream_) {
    if (!styles || !wsq || !dcoefs) return sgv_fail(SGV_ERR_INVALID_ARG, "demod_coefs: NULL pointer");
    if (n < 1 || oc < 1 || ic < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "demod_coefs: sizes must be positive");
    if (n > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "demod_coefs: batch is too large");
    hipStream_t stream = (hipStream_t)stream_;
    sgv_launch_scope scope(SGV_K_MODULATE, stream, ((double)oc * ic + (double)n * ic + (double)n * oc) * 4.0);
    hipLaunchKernelGGL(demod_coefs_kernel, dim3((oc + 3) / 4, n), dim3(256), 0, stream, styles, wsq, dcoefs, n, oc, ic, eps);
    return sgv_check_launch("demod_coefs_kernel");
}

// First-order backward of the demodulation coefficients (round 6: the chain was 8 torch launches per layer on [N, O] / [O, I]-sized tensors, ~100 per generator
// backward):   g[n,o] = -grad_d[n,o] d[n,o]^3;   grad_w[o,i,k] = W[o,i,k] sum_n g[n,o] s[n,i]^2;   grad_s[n,i] = s[n,i] sum_o g[n,o] q[o,i]
// (d = (s^2 q^T + eps)^(-1/2): dd/d(s^2 q^T) = -d^3 / 2, and the 2 of d(s^2) = 2 s ds, d(q) = 2 W dW cancels it).
// One launch: workgroups [0, O x ceil(I / 256)) own one output channel and 256 input channels of grad_w, the next N x ceil(I / 256) one sample of grad_s.
__global__ __launch_bounds__(256) void demod_backward_kernel(const float* __restrict__ grad_d, const float* __restrict__ d, const float* __restrict__ s, const float* __restrict__ q,
                                                            const float* __restrict__ w, float* __restrict__ grad_w, float* __restrict__ grad_s,
                                                            int n_samples, int oc, int ic, int kk, int w_blocks) {
    const int itiles = (ic + 255) / 256;
    if ((int)blockIdx.x < w_blocks) {
        const int o = blockIdx.x / itiles, i = (blockIdx.x % itiles) * 256 + threadIdx.x;
        if (i >= ic) return;
        float acc = 0.f;
        for (int n = 0; n < n_samples; n++) {
            const float dd = d[(size_t)n * oc + o];
            const float g = -grad_d[(size_t)n * oc + o] * dd * dd * dd;      // (wave-uniform: scalar loads)
            const float sv = s[(size_t)n * ic + i];
            acc = __builtin_fmaf(g, sv * sv, acc);
        }
        const size_t at = ((size_t)o * ic + i) * kk;
        for (int k = 0; k < kk; k++) grad_w[at + k] = w[at + k] * acc;
    } else {
        const int b = blockIdx.x - w_blocks;
        const int n = b / itiles, i = (b % itiles) * 256 + threadIdx.x;
        if (i >= ic) return;
        float acc = 0.f;
        for (int o = 0; o < oc; o++) {
            const float dd = d[(size_t)n * oc + o];
            const float g = -grad_d[(size_t)n * oc + o] * dd * dd * dd;
            acc = __builtin_fmaf(g, q[(size_t)o * ic + i], acc);
        }
        grad_s[(size_t)n * ic + i] = s[(size_t)n * ic + i] * acc;
    }
}

extern "C" int sgv_demod_coefs_backward(const float* grad_d, const float* dcoefs, const float* styles, const float* wsq, const float* weight, float* grad_w, float* grad_s,
                                        int32_t n, int32_t oc, int32_t ic, int32_t kk, void* stream_) {
    if (!grad_d || !dcoefs || !styles || !wsq || (!grad_w && !grad_s) || (grad_w && !weight)) return sgv_fail(SGV_ERR_INVALID_ARG, "demod_coefs_backward: NULL pointer");
    if (n < 1 || oc < 1 || ic < 1 || kk < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "demod_coefs_backward: sizes must be positive");
    hipStream_t stream = (hipStream_t)stream_;
    const int itiles = (ic + 255) / 256;
    const int64_t wb = grad_w ? (int64_t)oc * itiles : 0, sb = grad_s ? (int64_t)n * itiles : 0;
    if (wb + sb > 0x7fffffff) return sgv_fail(SGV_ERR_TOO_LARGE, "demod_coefs_backward: too many workgroups");
    sgv_launch_scope scope(SGV_K_MODULATE, stream, ((double)oc * ic * (2.0 * kk + 1.0) + 2.0 * n * ic + 2.0 * n * oc) * 4.0);
    hipLaunchKernelGGL(demod_backward_kernel, dim3((unsigned)(wb + sb)), dim3(256), 0, stream, grad_d, dcoefs, styles, wsq, weight, grad_w, grad_s, n, oc, ic, kk, (int)wb);
    return sgv_check_launch("demod_backward_kernel");
}

extern "C" int sgv_scale_channels(const void* x, const float* s, void* y, int32_t n, int32_t c, int32_t hw, int dtype,
                                  void* stream_) {
    if (!x || !s || !y) return sgv_fail(SGV_ERR_INVALID_ARG, "scale_channels: NULL pointer");
    if (n < 1 || c < 1 || hw < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "scale_channels: sizes must be positive");
    if ((int64_t)n * c * hw > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "scale_channels: x is too large");
    const size_t es = sgv_dtype_size(dtype);
    if (es == 0) return sgv_fail(SGV_ERR_UNSUPPORTED, "scale_channels: unknown dtype %d", dtype);
    hipStream_t stream = (hipStream_t)stream_;
    const int total = n * c * hw;
    sgv_launch_scope scope(SGV_K_MODULATE, stream, 2.0 * total * es + (double)n * c * 4.0);
    switch (dtype) {
        case SGV_F32: launch_scale<float>(x, s, y, total, hw, stream, scope); break;
        case SGV_F16: launch_scale<sgv_half_t>(x, s, y, total, hw, stream, scope); break;
        case SGV_BF16: launch_scale<sgv_bf16_t>(x, s, y, total, hw, stream, scope); break;
        case SGV_F64: return sgv_fail(SGV_ERR_UNSUPPORTED, "scale_channels: fp64 not supported");
    }
    return sgv_check_launch("scale_channels_kernel");
}
