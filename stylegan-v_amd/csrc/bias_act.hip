// bias_act for gfx950:  y = clamp(act(x + b) * gain)  and its 1st / 2nd derivative forms.
//
// Formulas follow the reference kernel (src/torch_utils/ops/bias_act.cu:39-146) operation by
// operation -- same order of bias add, activation, `y *= gain * dy`, clamp -- so linear / relu /
// lrelu (only exactly-rounded add/mul/select) are bit-identical to the scalar oracle.
//
// Design: pure HBM stream (AI ~0.4 flop/B).  One lane moves one 16-B vector per stream
// (dwordx4 for fp32, 8 x 16-bit for fp16/bf16, 2 x fp64); the grid covers the whole tensor.  The bias index (xi / step_b) % size_b (bias_act.cu:44) is evaluated once per
// 16-B vector when step_b is a multiple of the vector length (NCHW feature maps: step_b = H*W),
// as a vector load when step_b == 1 (fully-connected: bias runs along the fastest axis), and per
// element otherwise.  Activation and grad order are template parameters; the presence of each
// optional stream and the bias mode are wave-uniform scalar branches.
//
// Algorithmic bytes per launch: size_x * sizeof(T) * (#input streams + 1) + size_b * sizeof(T).

#include "sgv_common.h"

#pragma clang fp contract(off)

namespace {

struct ba_params {
    const void* x;
    const void* b;
    const void* xref;
    const void* yref;
    const void* dy;
    void* y;
    float alpha, gain, clamp;
    int size_x, size_b, step_b;
    float* db;     // optional: db[channel] += sum of the outputs (bias gradient), fp32 atomics
    int db_mode;   // 1: every wave lies inside one channel plane (one atomic per wave), 2: one atomic per lane
    int db_slots;  // db is [db_slots][size_b]: waves spread their atomics over the slots (power of two), the caller adds the slots up
};

template <typename S> __device__ __forceinline__ S s_exp(S v);
template <> __device__ __forceinline__ float s_exp<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double s_exp<double>(double v) { return exp(v); }
template <typename S> __device__ __forceinline__ S s_log(S v);
template <> __device__ __forceinline__ float s_log<float>(float v) { return logf(v); }
template <> __device__ __forceinline__ double s_log<double>(double v) { return log(v); }

// One element.  A = activation index (1..9), G = gradient order; mirrors bias_act.cu:51-142.
template <typename S, int A, int G>
__device__ __forceinline__ S ba_eval(S x, S b, S xref, S yref, S dy, S alpha, S gain, S clamp) {
    const S one = (S)1, two = (S)2, exp_range = (S)80, half_exp_range = (S)40;
    const S selu_scale = (S)1.0507009873554804934193349852946;
    const S selu_alpha = (S)1.6732632423543772848170429916717;
    S yy = (gain != 0) ? yref / gain : 0;
    S y = 0;
    if (G == 0) x += b; else xref += b;

    if (A == 1) { if (G == 0 || G == 1) y = x; }
    if (A == 2) { if (G == 0) y = (x > 0) ? x : 0; if (G == 1) y = (yy > 0) ? x : 0; }
    if (A == 3) { if (G == 0) y = (x > 0) ? x : x * alpha; if (G == 1) y = (yy > 0) ? x : x * alpha; }
    if (A == 4) {
        if (G == 0) { S c = s_exp<S>(x); S d = one / c; y = (x < -exp_range) ? -one : (x > exp_range) ? one : (c - d) / (c + d); }
        if (G == 1) y = x * (one - yy * yy);
        if (G == 2) y = x * (one - yy * yy) * (-two * yy);
    }
    if (A == 5) {
        if (G == 0) y = (x < -exp_range) ? 0 : one / (s_exp<S>(-x) + one);
        if (G == 1) y = x * yy * (one - yy);
        if (G == 2) y = x * yy * (one - yy) * (one - two * yy);
    }
    if (A == 6) {
        if (G == 0) y = (x >= 0) ? x : s_exp<S>(x) - one;
        if (G == 1) y = (yy >= 0) ? x : x * (yy + one);
        if (G == 2) y = (yy >= 0) ? 0 : x * (yy + one);
    }
    if (A == 7) {
        if (G == 0) y = (x >= 0) ? selu_scale * x : (selu_scale * selu_alpha) * (s_exp<S>(x) - one);
        if (G == 1) y = (yy >= 0) ? x * selu_scale : x * (yy + selu_scale * selu_alpha);
        if (G == 2) y = (yy >= 0) ? 0 : x * (yy + selu_scale * selu_alpha);
    }
    if (A == 8) {
        if (G == 0) y = (x > exp_range) ? x : s_log<S>(s_exp<S>(x) + one);
        if (G == 1) y = x * (one - s_exp<S>(-yy));
        if (G == 2) { S c = s_exp<S>(-yy); y = x * c * (one - c); }
    }
    if (A == 9) {
        if (G == 0) {
            y = (x < -exp_range) ? 0 : x / (s_exp<S>(-x) + one);
        } else {
            S c = s_exp<S>(xref);
            S d = c + one;
            if (G == 1) y = (xref > half_exp_range) ? x : x * c * (xref + d) / (d * d);
            else y = (xref > half_exp_range) ? 0 : x * c * (xref * (two - d) + two * d) / (d * d * d);
            yref = (xref < -exp_range) ? 0 : xref / (s_exp<S>(-xref) + one) * gain;
        }
    }
    y *= gain * dy;
    if (clamp >= 0) {
        if (G == 0) y = (y > -clamp & y < clamp) ? y : (y >= 0) ? clamp : -clamp;
        else y = (yref > -clamp & yref < clamp) ? y : 0;
    }
    return y;
}

// 16-byte vector of T.
template <typename T> struct vec16 {
    static constexpr int N = 16 / sizeof(T);
    T e[N];
} __attribute__((aligned(16)));

// bmode: 0 no bias, 1 one bias per vector (step_b % N == 0), 2 bias vector (step_b == 1,
// size_b % N == 0), 3 per element.  bmode and the presence of xref/yref/dy are wave-uniform runtime
// values (scalar branches); activation and grad order are template parameters.
template <typename T, int A, int G>
__global__ __launch_bounds__(256) void bias_act_kernel(ba_params p, int bmode, int nt_store) {
    typedef typename sgv_traits<T>::acc_t S;
    constexpr int N = vec16<T>::N;
    const S alpha = (S)p.alpha, gain = (S)p.gain, clamp = (S)p.clamp;
    const int nvec = p.size_x / N;
    const vec16<T>* xv = (const vec16<T>*)p.x;
    const vec16<T>* xrv = (const vec16<T>*)p.xref;
    const vec16<T>* yrv = (const vec16<T>*)p.yref;
    const vec16<T>* dyv = (const vec16<T>*)p.dy;
    vec16<T>* yv = (vec16<T>*)p.y;
    const T* bp = (const T*)p.b;
    const bool has_xref = (G > 0) && xrv != nullptr;
    const bool has_yref = (G > 0) && yrv != nullptr;
    const bool has_dy = dyv != nullptr;  // the reference multiplies by dy at every grad order (bias_act.cu:133)

    // One 16-B vector per lane and a grid that covers the tensor: on MI355X many short-lived workgroups sweeping a
    // compact window of memory stream HBM ~25 % faster than 2048 persistent grid-stride blocks (6.1 vs 4.9 TB/s on a
    // float4 copy of 1 GB).  The loop only runs more than once for tensors beyond 2^31 / 16 vectors per grid limit.
    for (int vi = blockIdx.x * blockDim.x + threadIdx.x; vi < nvec; vi += gridDim.x * blockDim.x) {
        vec16<T> vx = xv[vi], vxr = vx, vyr = vx, vdy = vx, vo;
        if (has_xref) vxr = xrv[vi];
        if (has_yref) vyr = yrv[vi];
        if (has_dy) vdy = dyv[vi];
        const int xi = vi * N;
        S bias[N];
#pragma unroll
        for (int k = 0; k < N; k++) bias[k] = 0;
        if (bmode == 1) {
            S b0 = sgv_traits<T>::load(bp + (xi / p.step_b) % p.size_b);
#pragma unroll
            for (int k = 0; k < N; k++) bias[k] = b0;
        } else if (bmode == 2) {
            vec16<T> vb = *(const vec16<T>*)(bp + xi % p.size_b);
#pragma unroll
            for (int k = 0; k < N; k++) bias[k] = sgv_traits<T>::load(&vb.e[k]);
        } else if (bmode == 3) {
#pragma unroll
            for (int k = 0; k < N; k++) bias[k] = sgv_traits<T>::load(bp + ((xi + k) / p.step_b) % p.size_b);
        }
#pragma unroll
        for (int k = 0; k < N; k++) {
            S x = sgv_traits<T>::load(&vx.e[k]);
            S xr = has_xref ? sgv_traits<T>::load(&vxr.e[k]) : (S)0;
            S yr = has_yref ? sgv_traits<T>::load(&vyr.e[k]) : (S)0;
            S dy = has_dy ? sgv_traits<T>::load(&vdy.e[k]) : (S)1;
            sgv_traits<T>::store(&vo.e[k], ba_eval<S, A, G>(x, bias[k], xr, yr, dy, alpha, gain, clamp));
        }
        if (nt_store) {
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(__builtin_bit_cast(u4, vo), (u4*)&yv[vi]);
        } else {
            yv[vi] = vo;
        }
        if (p.db) {   // bias gradient = per-channel sum of what was just written (bias_act.py:185 `dx.sum(...)` of the reference)
            float part = 0.f;
#pragma unroll
            for (int k = 0; k < N; k++) part += (float)sgv_traits<T>::load(&vo.e[k]);
            const int ch = (xi / p.step_b) % p.size_b + ((vi >> 6) & (p.db_slots - 1)) * p.size_b;
            if (p.db_mode == 1) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
                if ((threadIdx.x & 63) == 0) atomicAdd(p.db + ch, part);
            } else {
                atomicAdd(p.db + ch, part);
            }
        }
    }

    // Tail (size_x not a multiple of the vector length): handled by the first lanes of block 0.
    const int tail0 = nvec * N;
    if (blockIdx.x == 0 && (int)threadIdx.x < p.size_x - tail0) {
        const int xi = tail0 + threadIdx.x;
        S b = (bmode != 0) ? sgv_traits<T>::load(bp + (xi / p.step_b) % p.size_b) : (S)0;
        S x = sgv_traits<T>::load((const T*)p.x + xi);
        S xr = has_xref ? sgv_traits<T>::load((const T*)p.xref + xi) : (S)0;
        S yr = has_yref ? sgv_traits<T>::load((const T*)p.yref + xi) : (S)0;
        S dy = has_dy ? sgv_traits<T>::load((const T*)p.dy + xi) : (S)1;
        sgv_traits<T>::store((T*)p.y + xi, ba_eval<S, A, G>(x, b, xr, yr, dy, alpha, gain, clamp));
    }
}

typedef void (*ba_fn)(ba_params, int, int);

template <typename T, int A>
ba_fn pick_grad(int grad) {
    if (grad == 0) return bias_act_kernel<T, A, 0>;
    if (grad == 1) return bias_act_kernel<T, A, 1>;
    return bias_act_kernel<T, A, 2>;
}

template <typename T>
ba_fn pick_act(int act, int grad) {
    switch (act) {
        case 1: return pick_grad<T, 1>(grad);
        case 2: return pick_grad<T, 2>(grad);
        case 3: return pick_grad<T, 3>(grad);
        case 4: return pick_grad<T, 4>(grad);
        case 5: return pick_grad<T, 5>(grad);
        case 6: return pick_grad<T, 6>(grad);
        case 7: return pick_grad<T, 7>(grad);
        case 8: return pick_grad<T, 8>(grad);
        case 9: return pick_grad<T, 9>(grad);
        default: return nullptr;
    }
}

}  // namespace

static int bias_act_launch(const sgv_bias_act_params* p, float* db, int db_slots, int dtype, void* stream_) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "bias_act: params is NULL");
    const size_t es = sgv_dtype_size(dtype);
    if (es == 0) return sgv_fail(SGV_ERR_UNSUPPORTED, "bias_act: unknown dtype %d", dtype);
    if (!p->x || !p->y) return sgv_fail(SGV_ERR_INVALID_ARG, "bias_act: x and y must be non-NULL");
    if (p->size_x < 0) return sgv_fail(SGV_ERR_TOO_LARGE, "bias_act: x is too large");
    if (p->grad < 0) return sgv_fail(SGV_ERR_INVALID_ARG, "bias_act: grad must be non-negative");
    if (p->grad > 2) return sgv_fail(SGV_ERR_UNSUPPORTED, "bias_act: grad order %d not supported", p->grad);
    if (p->act < 1 || p->act > 9) return sgv_fail(SGV_ERR_UNSUPPORTED, "bias_act: no kernel found for the specified activation func (%d)", p->act);
    if (p->b && (p->size_b < 1 || p->step_b < 1)) return sgv_fail(SGV_ERR_INVALID_ARG, "bias_act: b has wrong number of elements");
    if (p->size_x == 0) return SGV_OK;
    hipStream_t stream = (hipStream_t)stream_;

    const int nvecel = (int)(16 / es);
    int bmode = 0;
    if (p->b) {
        if (p->step_b % nvecel == 0) bmode = 1;
        else if (p->step_b == 1 && p->size_b % nvecel == 0 && ((uintptr_t)p->b % 16) == 0) bmode = 2;
        else bmode = 3;
    }
    const bool xr = p->xref != nullptr, yr = p->yref != nullptr, dy = p->dy != nullptr;
    // 16-B vector access needs 16-B aligned bases (torch allocations are 512-B aligned; views with
    // a storage offset may not be).
    const uintptr_t align_or = (uintptr_t)p->x | (uintptr_t)p->y | (uintptr_t)p->xref | (uintptr_t)p->yref | (uintptr_t)p->dy;
    if (align_or % 16 != 0) return sgv_fail(SGV_ERR_INVALID_ARG, "bias_act: tensor base pointers must be 16-byte aligned");

    ba_fn fn = nullptr;
    switch (dtype) {
        case SGV_F32: fn = pick_act<float>(p->act, p->grad); break;
        case SGV_F16: fn = pick_act<sgv_half_t>(p->act, p->grad); break;
        case SGV_BF16: fn = pick_act<sgv_bf16_t>(p->act, p->grad); break;
        case SGV_F64: fn = pick_act<double>(p->act, p->grad); break;
    }
    if (!fn) return sgv_fail(SGV_ERR_UNSUPPORTED, "bias_act: no kernel for act=%d grad=%d", p->act, p->grad);

    ba_params kp;
    kp.x = p->x; kp.b = p->b; kp.xref = p->xref; kp.yref = p->yref; kp.dy = p->dy; kp.y = p->y;
    kp.alpha = p->alpha; kp.gain = p->gain; kp.clamp = p->clamp;
    kp.size_x = p->size_x; kp.size_b = p->b ? p->size_b : 1; kp.step_b = p->b ? p->step_b : 1;
    kp.db = db; kp.db_mode = 0; kp.db_slots = db_slots;
    if (db) {
        // the fused bias gradient needs one channel per 16-B vector and no scalar tail
        if (!p->b || bmode != 1 || p->size_x % nvecel != 0) return sgv_fail(SGV_ERR_UNSUPPORTED, "bias_act_db: needs a bias with step_b %% %d == 0 and size_x %% %d == 0", nvecel, nvecel);
        if (p->grad != 1) return sgv_fail(SGV_ERR_INVALID_ARG, "bias_act_db: only the grad = 1 form produces a bias gradient");
        kp.db_mode = (p->step_b % (64 * nvecel) == 0) ? 1 : 2;
    }

    const int nvec = p->size_x / nvecel;
    int blocks = (nvec + 255) / 256;
    if (blocks < 1) blocks = 1;
    // outputs larger than the 256 MiB Infinity Cache cannot be re-read from cache by the next op: stream them past it
    const int nt_store = ((double)p->size_x * es > 300e6) ? 1 : 0;
    const int streams = 2 + (xr ? 1 : 0) + (yr ? 1 : 0) + (dy ? 1 : 0);
    const double bytes = (double)p->size_x * es * streams + (p->b ? (double)p->size_b * es : 0.0);
    sgv_launch_scope scope(SGV_K_BIAS_ACT, stream, bytes);
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, stream, kp, bmode, nt_store);
    sgv_note_variant(SGV_V_bias_act);
    return sgv_check_launch("bias_act_kernel");
}

extern "C" int sgv_bias_act(const sgv_bias_act_params* p, int dtype, void* stream) { return bias_act_launch(p, nullptr, 1, dtype, stream); }

// grad = 1 form that also accumulates the bias gradient: db[slot][c] += partial sums of the result over channel c (fp32 atomics spread over
// `db_slots` copies to keep same-address contention low; the caller zero-initialises db[db_slots][size_b] and adds the slots up).
extern "C" int sgv_bias_act_db(const sgv_bias_act_params* p, float* db, int db_slots, int dtype, void* stream) {
    if (!db) return sgv_fail(SGV_ERR_INVALID_ARG, "bias_act_db: db is NULL");
    if (db_slots < 1 || (db_slots & (db_slots - 1))) return sgv_fail(SGV_ERR_INVALID_ARG, "bias_act_db: db_slots must be a power of two");
    return bias_act_launch(p, db, db_slots, dtype, stream);
}
