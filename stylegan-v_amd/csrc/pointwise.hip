// 1x1 convolutions with a tiny channel count on one side (ToRGB: Cin -> 3, fromRGB: 3 -> C) for gfx950.
//
// Reference: the 1x1 `conv2d` of ToRGBLayer (src/training/networks.py:148-163, modulated, C_out = img_channels = 3) and of the
// discriminator's `fromrgb` Conv2dLayer (networks.py:447, C_in = 3), which the reference hands to cuDNN through
// conv2d_resample.py:40-54.  With 3 channels on one side the arithmetic intensity is ~1.4 flop/B: these
// are HBM streams, not GEMMs -- MIOpen runs them at 1.8 TFLOP/s (1.3 ms for [96,64,256,256]) where moving the bytes takes
// 0.3 ms.  Three kernels, NCHW, fp32 accumulate, weights fp32 and optionally per sample (ToRGB folds the style into them:
// w[n,o,i] = W[o,i] * s[n,i], so the separate x*s pass disappears as well):
//
//   pw_many2few   y[n,f,p] = sum_m w[n,f,m] * x[n,m,p]      F <= 4 outputs; a lane owns 4 pixels (16 B), walks the M input
//                                                           planes with one coalesced 1-KiB-per-wave load each
//   pw_few2many   y[n,m,p] = sum_f w[n,m,f] * x[n,f,p]      F <= 4 inputs held in registers, one 16-B store per output plane
//   pw_outer      out[n,f,m] += sum_p a[n,f,p] * b[n,m,p]   weight-gradient reduction: the F-side pixels stay in registers,
//                                                           the M planes stream past, DPP/shuffle wave reduction, one atomic
//                                                           per (wave, f, m)
//
// Algorithmic bytes: (M + F) * N * HW * sizeof(T) for each of them (+ the weights, negligible).

#include "sgv_common.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace {

constexpr int FMAX = 4;

template <typename T> struct px4 { T e[4]; } __attribute__((aligned(sizeof(T) * 4)));

template <typename T> __device__ __forceinline__ void load4(const T* p, float* v) {
    px4<T> q = *(const px4<T>*)p;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = sgv_traits<T>::load(&q.e[i]);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v) {
    px4<T> q;
#pragma unroll
    for (int i = 0; i < 4; i++) sgv_traits<T>::store(&q.e[i], v[i]);
    *(px4<T>*)p = q;
}

// bias_act's first derivative for the two activations of this path, from the saved OUTPUT y (bias_act.cu:51-142 with yref): linear / lrelu slope
// times gain, zero where the output sat on the clamp
__device__ __forceinline__ float pw_act_grad(float dy, float y, int act, float alpha, float gain, float clamp) {
    float dz = ((act == 3 && !(y > 0.f)) ? dy * alpha : dy) * gain;
    if (clamp >= 0.f && !(y > -clamp & y < clamp)) dz = 0.f;
    return dz;
}

struct pw_params {
    const void* x;
    const float* w;
    void* y;
    int n, cm, cf, hw;       // hw is a multiple of 4 on this path
    int64_t w_stride_n;      // 0: weights shared by the batch
    int m_per_z;             // few2many: planes of the many side per blockIdx.z slice
    // few2many epilogue (sgv_pointwise_act): y = clamp(act(y + bias[m]) * gain) in bias_act.hip's own operation order; act 0 = none
    const float* bias;
    int act;
    float alpha, gain, clamp;
    // many2few prologue (sgv_pointwise_small_gradin): x is a gradient dy and is turned into pw_act_grad(dy, yref) on its way in
    const void* yref;
    float* y_amax;           // few2many with epilogue, fp32: max |y| as a by-product (sgv_amax_sink: fromRGB's output feeds a 3x3 convolution), or NULL
};

// MS == 1: grid = (ceil(hw/4/256), n), lane -> pixel quad, every lane walks all M planes.
// MS == 4 (small images: too few quads to fill the chip): grid = (ceil(hw/4/64), n); the 4 waves of a workgroup share 64
// quads and take a quarter of the M planes each, partial sums meet in LDS.
template <typename T, int F, int MS>
__global__ __launch_bounds__(256) void pw_many2few_kernel(pw_params p) {
    const int n = blockIdx.y;
    const int wave = threadIdx.x >> 6;
    const int q = MS == 1 ? blockIdx.x * 256 + threadIdx.x : blockIdx.x * 64 + (threadIdx.x & 63);
    const bool live = q * 4 < p.hw;
    if (MS == 1 && !live) return;
    const int m_per = (p.cm + MS - 1) / MS;
    const int m0 = MS == 1 ? 0 : __builtin_amdgcn_readfirstlane(wave) * m_per;
    const int m1 = MS == 1 ? p.cm : min(p.cm, m0 + m_per);
    const T* x = (const T*)p.x + (size_t)n * p.cm * p.hw + (size_t)(live ? q : 0) * 4;
    const float* w = p.w + n * p.w_stride_n;   // [F][cm], wave-uniform -> scalar loads
    float acc[F][4];
#pragma unroll
    for (int f = 0; f < F; f++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[f][i] = 0.f;
#pragma unroll 4
    for (int m = m0; m < m1; m++) {
        float v[4];
        load4<T>(x + (size_t)m * p.hw, v);
        if (p.yref) {
            float yv[4];
            load4<T>((const T*)p.yref + (x - (const T*)p.x) + (size_t)m * p.hw, yv);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = pw_act_grad(v[i], yv[i], p.act, p.alpha, p.gain, p.clamp);
        }
#pragma unroll
        for (int f = 0; f < F; f++) {
            const float wf = w[f * p.cm + m];
#pragma unroll
            for (int i = 0; i < 4; i++) acc[f][i] = __builtin_fmaf(v[i], wf, acc[f][i]);
        }
    }
    if (MS > 1) {
        __shared__ float part[MS - 1 > 0 ? MS - 1 : 1][F][4][64];
        const int lane = threadIdx.x & 63;
        if (wave > 0) {
#pragma unroll
            for (int f = 0; f < F; f++)
#pragma unroll
                for (int i = 0; i < 4; i++) part[wave - 1][f][i][lane] = acc[f][i];
        }
        __syncthreads();
        if (wave > 0 || !live) return;
#pragma unroll
        for (int f = 0; f < F; f++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float t = acc[f][i];
#pragma unroll
                for (int k = 0; k < MS - 1; k++) t += part[k][f][i][lane];   // fixed order: results do not depend on scheduling
                acc[f][i] = t;
            }
    }
    T* y = (T*)p.y + (size_t)n * F * p.hw + (size_t)q * 4;
#pragma unroll
    for (int f = 0; f < F; f++) store4<T>(y + (size_t)f * p.hw, acc[f]);
}

template <typename T, int F, int EPI = 0>
__global__ __launch_bounds__(256) void pw_few2many_kernel(pw_params p) {
    const int n = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned amx = 0u;
    if (q * 4 < p.hw) {
        const T* x = (const T*)p.x + (size_t)n * F * p.hw + (size_t)q * 4;
        const float* w = p.w + n * p.w_stride_n;   // [cm][F]
        float v[F][4];
#pragma unroll
        for (int f = 0; f < F; f++) load4<T>(x + (size_t)f * p.hw, v[f]);
        T* y = (T*)p.y + (size_t)n * p.cm * p.hw + (size_t)q * 4;
        const int mz0 = blockIdx.z * p.m_per_z, mz1 = min(p.cm, mz0 + p.m_per_z);
#pragma unroll 4
        for (int m = mz0; m < mz1; m++) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < F; f++) {
                const float wf = w[m * F + f];
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = __builtin_fmaf(v[f][i], wf, o[i]);
            }
            if (EPI) {
                const float bm = p.bias ? p.bias[m] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float t = o[i] + bm;
                    if (p.act == 3) t = (t > 0.f) ? t : t * p.alpha;
                    t *= p.gain;
                    if (p.clamp >= 0.f) t = (t > -p.clamp & t < p.clamp) ? t : (t >= 0.f) ? p.clamp : -p.clamp;
                    o[i] = t;
                    if constexpr (sizeof(T) == 4) amx = sgv_amax_fold(amx, t);
                }
            }
            store4<T>(y + (size_t)m * p.hw, o);
        }
    }
    // every lane of the wave arrives here (lanes beyond the image carry 0): the wave reduction needs them all
    if constexpr (EPI && sizeof(T) == 4) { if (p.y_amax) sgv_amax_commit(amx, p.y_amax); }
}

struct outer_params {
    const void* a;   // [n, F, hw]   (few side)
    const void* b;   // [n, cm, hw]  (many side)
    float* out;      // [n, F, cm] fp32, accumulated with atomics (zero-initialised by the caller)
    int n, cm, cf, hw;
    int chunk;       // pixels per workgroup (multiple of 1024)
    int use_lds;     // reduce the 4 waves of a workgroup through LDS before the atomics
    int m_per_z;     // planes of the many side per blockIdx.z slice
    // sgv_pointwise_outer_act: b is a gradient dy that becomes pw_act_grad(dy, yref); with ones_row the last few-side plane is the constant 1
    // (not in memory: a has F - 1 planes), so that out[n, F-1, m] = sum_p dz = the bias gradient
    const void* yref;
    int act;
    float alpha, gain, clamp;
    int ones_row;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// grid = (ceil(hw / chunk), n); each lane keeps PQ pixel quads of the F few-side planes in registers.
template <typename T, int F, int PQ>
__global__ __launch_bounds__(256) void pw_outer_kernel(outer_params p) {
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * p.chunk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    extern __shared__ float red[];   // [4 waves][F][cm] partial sums when p.use_lds
    const T* a = (const T*)p.a + (size_t)n * (p.ones_row ? F - 1 : F) * p.hw;
    const T* b = (const T*)p.b + (size_t)n * p.cm * p.hw;
    const T* yr = p.yref ? (const T*)p.yref + (size_t)n * p.cm * p.hw : nullptr;
    float av[F][PQ][4];
    int pix[PQ];
#pragma unroll
    for (int k = 0; k < PQ; k++) {
        pix[k] = p0 + (k * 256 + threadIdx.x) * 4;
#pragma unroll
        for (int f = 0; f < F; f++) {
            const bool in = pix[k] < p.hw && pix[k] < p0 + p.chunk;
            if (in && p.ones_row && f == F - 1) { av[f][k][0] = av[f][k][1] = av[f][k][2] = av[f][k][3] = 1.f; }
            else if (in) load4<T>(a + (size_t)f * p.hw + pix[k], av[f][k]);
            else { av[f][k][0] = av[f][k][1] = av[f][k][2] = av[f][k][3] = 0.f; }
        }
    }
    const int mz0 = blockIdx.z * p.m_per_z, mz1 = min(p.cm, mz0 + p.m_per_z), mzn = mz1 - mz0;
    for (int m = mz0; m < mz1; m++) {
        float s[F];
#pragma unroll
        for (int f = 0; f < F; f++) s[f] = 0.f;
#pragma unroll
        for (int k = 0; k < PQ; k++) {
            if (pix[k] < p.hw && pix[k] < p0 + p.chunk) {
                float bv[4];
                load4<T>(b + (size_t)m * p.hw + pix[k], bv);
                if (yr) {
                    float yv[4];
                    load4<T>(yr + (size_t)m * p.hw + pix[k], yv);
#pragma unroll
                    for (int i = 0; i < 4; i++) bv[i] = pw_act_grad(bv[i], yv[i], p.act, p.alpha, p.gain, p.clamp);
                }
#pragma unroll
                for (int f = 0; f < F; f++)
#pragma unroll
                    for (int i = 0; i < 4; i++) s[f] = __builtin_fmaf(av[f][k][i], bv[i], s[f]);
            }
        }
#pragma unroll
        for (int f = 0; f < F; f++) {
            const float t = wave_sum(s[f]);
            if (lane == 0) {
                if (p.use_lds) red[(wave * F + f) * p.m_per_z + (m - mz0)] = t;
                else atomicAdd(p.out + ((size_t)n * F + f) * p.cm + m, t);
            }
        }
    }
    if (p.use_lds) {   // one atomic per (workgroup, f, m) instead of one per wave
        __syncthreads();
        const int fm = F * p.m_per_z;
        for (int i = threadIdx.x; i < fm; i += 256) {
            const int f = i / p.m_per_z, mm = i - f * p.m_per_z;
            if (mm < mzn) atomicAdd(p.out + ((size_t)n * F + f) * p.cm + mz0 + mm, (red[i] + red[fm + i]) + (red[2 * fm + i] + red[3 * fm + i]));
        }
    }
}

// Slices of the many side so that a small image still launches >= ~2048 workgroups (256 CUs x 8), at least 8 planes per slice.
int m_slices(int wgs, int cm) {
    int z = (2048 + wgs - 1) / wgs;
    z = std::min(z, std::max(1, cm / 8));
    return std::max(1, std::min(z, 64));
}

template <typename T>
int launch_pw(int kind, pw_params pp, hipStream_t stream) {
    const int quads = pp.hw / 4;
    dim3 grid((unsigned)((quads + 255) / 256), (unsigned)pp.n);
    const bool split = kind == 0 && (int64_t)grid.x * pp.n < 1024 && pp.cm >= 16;
    if (split) grid.x = (unsigned)((quads + 63) / 64);
    if (kind == 1) {
        const int z = m_slices((int)(grid.x * grid.y), pp.cm);
        pp.m_per_z = (pp.cm + z - 1) / z;
        grid.z = (unsigned)((pp.cm + pp.m_per_z - 1) / pp.m_per_z);
    }
#define SGV_PW(F)                                                                                                   \
    if (pp.cf == F) {                                                                                               \
        if (kind == 0 && split) hipLaunchKernelGGL((pw_many2few_kernel<T, F, 4>), grid, dim3(256), 0, stream, pp);  \
        else if (kind == 0) hipLaunchKernelGGL((pw_many2few_kernel<T, F, 1>), grid, dim3(256), 0, stream, pp);      \
        else if (pp.act) hipLaunchKernelGGL((pw_few2many_kernel<T, F, 1>), grid, dim3(256), 0, stream, pp);         \
        else hipLaunchKernelGGL((pw_few2many_kernel<T, F>), grid, dim3(256), 0, stream, pp);                        \
        sgv_note_variant(kind == 0 ? SGV_V_pw_many2few : pp.act ? SGV_V_pw_few2many_act : SGV_V_pw_few2many);            \
        return SGV_OK;                                                                                              \
    }
    SGV_PW(1) SGV_PW(2) SGV_PW(3) SGV_PW(4)
#undef SGV_PW
    return sgv_fail(SGV_ERR_UNSUPPORTED, "pointwise: the small channel count must be 1..4");
}

template <typename T>
int launch_outer(outer_params op, hipStream_t stream) {
    dim3 grid((unsigned)((op.hw + op.chunk - 1) / op.chunk), (unsigned)op.n);
    const int z = m_slices((int)(grid.x * grid.y), op.cm);
    op.m_per_z = (op.cm + z - 1) / z;
    grid.z = (unsigned)((op.cm + op.m_per_z - 1) / op.m_per_z);
    op.use_lds = 16 * op.cf * op.m_per_z <= 32768 ? 1 : 0;
#define SGV_OUT(F)                                                                                                  \
    if (op.cf == F) { hipLaunchKernelGGL((pw_outer_kernel<T, F, 4>), grid, dim3(256), op.use_lds ? 16u * F * op.m_per_z : 0u, stream, op); sgv_note_variant(SGV_V_pw_outer); return SGV_OK; }
    SGV_OUT(1) SGV_OUT(2) SGV_OUT(3) SGV_OUT(4)
#undef SGV_OUT
    return sgv_fail(SGV_ERR_UNSUPPORTED, "pointwise: the small channel count must be 1..4");
}

int check_common(const void* x, const void* w, const void* y, int n, int cm, int cf, int hw, int dtype, const char* what) {
    if (!x || !w || !y) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: NULL pointer", what);
    if (n < 1 || cm < 1 || cf < 1 || cf > FMAX || hw < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: bad sizes", what);
    if (n > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "%s: batch is too large", what);
    if (hw % 4 != 0) return sgv_fail(SGV_ERR_UNSUPPORTED, "%s: H*W must be a multiple of 4", what);
    if ((int64_t)n * cm * hw > INT32_MAX) return sgv_fail(SGV_ERR_TOO_LARGE, "%s: tensor is too large", what);
    if (dtype == SGV_F64 || sgv_dtype_size(dtype) == 0) return sgv_fail(SGV_ERR_UNSUPPORTED, "%s: unsupported dtype %d", what, dtype);
    const size_t align = sgv_dtype_size(dtype) * 4;
    if (((uintptr_t)x | (uintptr_t)y) % align != 0) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: tensors must be aligned to 4 elements", what);
    return SGV_OK;
}

}  // namespace

extern "C" int sgv_pointwise_small(const sgv_pointwise_params* p, int dtype, void* stream_) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise: params is NULL");
    int rc = check_common(p->x, p->w, p->y, p->n, p->c_many, p->c_few, p->hw, dtype, "pointwise");
    if (rc != SGV_OK) return rc;
    if (p->kind != 0 && p->kind != 1) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise: kind must be 0 (many->few) or 1 (few->many)");
    hipStream_t stream = (hipStream_t)stream_;
    pw_params pp{p->x, p->w, p->y, p->n, p->c_many, p->c_few, p->hw, p->w_stride_n, p->c_many, nullptr, 0, 0.f, 1.f, -1.f, nullptr};
    const double bytes = (double)(p->c_many + p->c_few) * p->n * p->hw * sgv_dtype_size(dtype);
    sgv_launch_scope scope(SGV_K_POINTWISE, stream, bytes, 2.0 * p->c_many * p->c_few * (double)p->n * p->hw);
    switch (dtype) {
        case SGV_F32: rc = launch_pw<float>(p->kind, pp, stream); break;
        case SGV_F16: rc = launch_pw<sgv_half_t>(p->kind, pp, stream); break;
        default: rc = launch_pw<sgv_bf16_t>(p->kind, pp, stream); break;
    }
    if (rc != SGV_OK) return rc;
    return sgv_check_launch("pointwise kernel");
}

extern "C" int sgv_pointwise_act(const sgv_pointwise_params* p, const float* bias, int32_t act, float alpha, float gain, float clamp, int dtype, void* stream_) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_act: params is NULL");
    int rc = check_common(p->x, p->w, p->y, p->n, p->c_many, p->c_few, p->hw, dtype, "pointwise_act");
    if (rc != SGV_OK) return rc;
    if (p->kind != 1) return sgv_fail(SGV_ERR_UNSUPPORTED, "pointwise_act: only the few -> many form (kind 1) has an epilogue");
    if (dtype != SGV_F32) return sgv_fail(SGV_ERR_UNSUPPORTED, "pointwise_act: fp32 only (a 16-bit composition rounds between the two steps)");
    if (act != 1 && act != 3) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_act: act must be 1 (linear) or 3 (lrelu)");
    hipStream_t stream = (hipStream_t)stream_;
    pw_params pp{p->x, p->w, p->y, p->n, p->c_many, p->c_few, p->hw, p->w_stride_n, p->c_many, bias, act, alpha, gain, clamp, nullptr, nullptr};
    const double bytes = (double)(p->c_many + p->c_few) * p->n * p->hw * 4.0;
    sgv_launch_scope scope(SGV_K_POINTWISE, stream, bytes, 2.0 * p->c_many * p->c_few * (double)p->n * p->hw);
    pp.y_amax = scope.take_amax_sink();
    rc = launch_pw<float>(1, pp, stream);
    if (rc != SGV_OK) return rc;
    return sgv_check_launch("pointwise kernel");
}

extern "C" int sgv_pointwise_outer(const void* a_few, const void* b_many, float* out, int32_t n, int32_t c_few, int32_t c_many, int32_t hw,
                                   int dtype, void* stream_) {
    int rc = check_common(a_few, b_many, out, n, c_many, c_few, hw, dtype, "pointwise_outer");
    if (rc != SGV_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    outer_params op{a_few, b_many, out, n, c_many, c_few, hw, 4096, 0, c_many, nullptr, 0, 0.f, 1.f, -1.f, 0};
    const double bytes = (double)(c_many + c_few) * n * hw * sgv_dtype_size(dtype);
    sgv_launch_scope scope(SGV_K_POINTWISE, stream, bytes, 2.0 * c_many * c_few * (double)n * hw);
    switch (dtype) {
        case SGV_F32: rc = launch_outer<float>(op, stream); break;
        case SGV_F16: rc = launch_outer<sgv_half_t>(op, stream); break;
        default: rc = launch_outer<sgv_bf16_t>(op, stream); break;
    }
    if (rc != SGV_OK) return rc;
    return sgv_check_launch("pointwise_outer kernel");
}

extern "C" int sgv_pointwise_small_gradin(const sgv_pointwise_params* p, const void* yref, int32_t act, float alpha, float gain, float clamp, int dtype, void* stream_) {
    if (!p || !yref) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_gradin: NULL pointer");
    int rc = check_common(p->x, p->w, p->y, p->n, p->c_many, p->c_few, p->hw, dtype, "pointwise_gradin");
    if (rc != SGV_OK) return rc;
    if (p->kind != 0 || dtype != SGV_F32) return sgv_fail(SGV_ERR_UNSUPPORTED, "pointwise_gradin: the many -> few form (kind 0) on fp32 tensors only");
    if (act != 1 && act != 3) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_gradin: act must be 1 (linear) or 3 (lrelu)");
    if (((uintptr_t)yref) % 16 != 0) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_gradin: yref must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    pw_params pp{p->x, p->w, p->y, p->n, p->c_many, p->c_few, p->hw, p->w_stride_n, p->c_many, nullptr, act, alpha, gain, clamp, yref};
    sgv_launch_scope scope(SGV_K_POINTWISE, stream, (double)(2 * p->c_many + p->c_few) * p->n * p->hw * 4.0, 2.0 * p->c_many * p->c_few * (double)p->n * p->hw);
    rc = launch_pw<float>(0, pp, stream);
    if (rc != SGV_OK) return rc;
    return sgv_check_launch("pointwise kernel");
}

extern "C" int sgv_pointwise_outer_act(const void* a_few, const void* dy_many, const void* yref, float* out, int32_t n, int32_t c_few, int32_t c_many, int32_t hw,
                                       int32_t ones_row, int32_t act, float alpha, float gain, float clamp, int dtype, void* stream_) {
    if (!yref) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_outer_act: NULL pointer");
    int rc = check_common(a_few, dy_many, out, n, c_many, c_few, hw, dtype, "pointwise_outer_act");
    if (rc != SGV_OK) return rc;
    if (dtype != SGV_F32) return sgv_fail(SGV_ERR_UNSUPPORTED, "pointwise_outer_act: fp32 tensors only");
    if (act != 1 && act != 3) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_outer_act: act must be 1 (linear) or 3 (lrelu)");
    if (ones_row && c_few < 2) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_outer_act: ones_row needs c_few >= 2 (the constant plane is counted in c_few)");
    if ((((uintptr_t)yref) | (uintptr_t)dy_many) % 16 != 0) return sgv_fail(SGV_ERR_INVALID_ARG, "pointwise_outer_act: tensors must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    outer_params op{a_few, dy_many, out, n, c_many, c_few, hw, 4096, 0, c_many, yref, act, alpha, gain, clamp, ones_row ? 1 : 0};
    sgv_launch_scope scope(SGV_K_POINTWISE, stream, (double)(2 * c_many + c_few) * n * hw * 4.0, 2.0 * c_many * c_few * (double)n * hw);
    rc = launch_outer<float>(op, stream);
    if (rc != SGV_OK) return rc;
    return sgv_check_launch("pointwise_outer kernel");
}
