// Small-M dense layer on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32), with the element-wise steps of
// FullyConnectedLayer folded in.
//
// Reference: `FullyConnectedLayer.forward` (src/training/layers.py:108-138: w = weight * weight_gain, b = bias * bias_gain,
// addmm / matmul + bias_act), the MappingNetwork chain (layers.py:22-104: normalize_2nd_moment -> fc -> lrelu -> fc -> lrelu), the 26
// style affines of the synthesis network, the discriminator epilogue and the temporal-encoder heads.  Every one of them has M = a few
// dozen rows (videos or frames of one rank) against 64..8192 features: the 128x128-tile GEMM of gemm_kernel.h runs them at a
// fraction of a TFLOP/s on 4 workgroups, and the vendor GEMM needs a second launch for bias + activation.  These problems are
// bound by streaming the weight once; what a kernel can win is launches and passes:
//
//   sgv_fc   C[m,n] = epi( sum_k A(m,k) * B(k,n) ),  A / B addressed by (row stride, column stride) so that one kernel serves
//            forward   y  = act(rownorm(x) @ (W * wg)^T + b * bg) * gain        A = x,  B = W^T     prologue: row RMS normalisation of x
//            data grad dx = (dz @ W) * wg                                      A = dz, B = W       (mapping input), operand A can be
//            weight gr dW = (dz^T @ x) * wg,  db = bg * sum_m dz               A = dz^T, B = x     "dy with the activation gradient
//                                                                                                   applied from the saved output"
// Tiling: a workgroup owns a 32 x 32 output tile (grid = N/32 x M/32: 16..256 workgroups for the layers above); its waves -- one per 8-deep
// k step, at most sixteen -- split K, each accumulates whole 32x32 MFMA tiles (two independent accumulators), partial tiles are added through
// LDS, then the epilogue.  (The wave count follows K: the weight gradient of the 8192 -> 512 epilogue layer is 4,096 tiles of K = 32; with
// sixteen waves each of them spent its time adding twelve empty partial tiles: 86 -> 19 us.)
// Algorithmic bytes: 4 * (M*K + N*K + M*N); flops 2*M*N*K.

#include "sgv_common.h"

#include <algorithm>
#include <stdlib.h>

namespace sgv_fck {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct fc_params {
    const float* a; int64_t sam, sak;     // A(m,k) = a[m * sam + k * sak]
    const float* aref;                    // optional, same indexing as a: A(m,k) *= act'(aref(m,k)) * gain  (bias_act gradient from the output)
    const float* b; int64_t sbk, sbn;     // B(k,n) = b[k * sbk + n * sbn]
    float* c; int64_t scm, scn;           // C(m,n) -> c[m * scm + n * scn]
    const float* bias;                    // [n] or NULL
    float* colsum;                        // optional [m] (of C's rows = A's rows): colsum[m] = bias_gain * sum_k A(m,k)   (bias gradient in the dW form)
    int m, n, k;
    int normalize;                        // forward prologue: A rows scaled by rsqrt(mean_k A^2 + 1e-8)  (normalize_2nd_moment)
    int act;                              // 1 linear, 3 lrelu: epilogue activation (forward) / activation of aref (gradient forms)
    float alpha, gain;                    // of that activation
    float wgain, bgain;                   // C = acc * wgain + bias * bgain
    int epilogue_act;                     // apply act/gain in the epilogue (forward form)
    int64_t sab, sbb, scb;                // batch strides (blockIdx.z)
    int accumulate;                       // C += ...
};

// AK / BK: operand is contiguous along k (row-major x, W in the forward form: each lane streams its own row with 16-B loads) or along the
// tile's row / column index (consecutive lanes read consecutive floats: scalar loads, one 128-B line per half wave).
// WAVES split K; every wave keeps two accumulators so that consecutive MFMAs are independent (a v_mfma_f32_32x32x2_f32 takes 64 cycles
// and a dependent one cannot start earlier): the serial chain of the K = 8192 epilogue layer is 128 MFMAs instead of 1024.
constexpr int FC_WAVES = 16;     // at most

// FC_UNROLL = k steps whose loads are issued together
template <int AK, int BK, int FC_UNROLL>
__device__ __forceinline__ void fc_body(const fc_params& p, const size_t bz) {
    extern __shared__ float red[];          // [max(waves - 1, 1)][32 * 32]
    __shared__ float rowstat[32];
    const int waves = blockDim.x >> 6;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, kk = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int am = m0 + r, bn = n0 + r;
    const bool a_ok = am < p.m, b_ok = bn < p.n;
    const float* ap = p.a + bz * p.sab + (size_t)(a_ok ? am : 0) * p.sam;
    const float* arp = p.aref ? p.aref + bz * p.sab + (size_t)(a_ok ? am : 0) * p.sam : nullptr;
    const float* bp = p.b + bz * p.sbb + (size_t)(b_ok ? bn : 0) * p.sbn;
    if (t < 32) rowstat[t] = 0.f;

    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; e++) { acc0[e] = 0.f; acc1[e] = 0.f; }
    float a_sum = 0.f, a_sq = 0.f;
    // a step covers 8 consecutive k: lane (r, kk) holds k = k0 + 4 * kk + j for j = 0..3, MFMA j contracts the pair (k0 + j, k0 + 4 + j)
    const int steps = (p.k + 7) / 8;
    const bool vec_a = AK && (p.k % 4 == 0) && ((((uintptr_t)p.a) | (uintptr_t)(p.sam * 4) | (uintptr_t)(p.sab * 4)) % 16 == 0) && (!p.aref || ((uintptr_t)p.aref % 16 == 0));
    const bool vec_b = BK && (p.k % 4 == 0) && ((((uintptr_t)p.b) | (uintptr_t)(p.sbn * 4) | (uintptr_t)(p.sbb * 4)) % 16 == 0);
    for (int i0 = wave; i0 < steps; i0 += waves * FC_UNROLL) {
        float av[FC_UNROLL][4], bv[FC_UNROLL][4], yv[FC_UNROLL][4];
        // all loads of FC_UNROLL steps first (a step past the end loads nothing and contributes zeros) ...
#pragma unroll
        for (int u = 0; u < FC_UNROLL; u++) {
            const int k0 = 8 * (i0 + u * waves) + 4 * kk;
            if (vec_a) {
                const bool ok = a_ok && k0 < p.k;
                const float4 q = ok ? *(const float4*)(ap + k0) : float4{0.f, 0.f, 0.f, 0.f};
                av[u][0] = q.x; av[u][1] = q.y; av[u][2] = q.z; av[u][3] = q.w;
                if (arp) { const float4 w = ok ? *(const float4*)(arp + k0) : float4{0.f, 0.f, 0.f, 0.f}; yv[u][0] = w.x; yv[u][1] = w.y; yv[u][2] = w.z; yv[u][3] = w.w; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool ok = a_ok && k0 + j < p.k;
                    av[u][j] = ok ? ap[(size_t)(k0 + j) * p.sak] : 0.f;
                    if (arp) yv[u][j] = ok ? arp[(size_t)(k0 + j) * p.sak] : 0.f;
                }
            }
            if (vec_b) {
                const float4 q = (b_ok && k0 < p.k) ? *(const float4*)(bp + k0) : float4{0.f, 0.f, 0.f, 0.f};
                bv[u][0] = q.x; bv[u][1] = q.y; bv[u][2] = q.z; bv[u][3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) bv[u][j] = (b_ok && k0 + j < p.k) ? bp[(size_t)(k0 + j) * p.sbk] : 0.f;
            }
        }
        // ... then the arithmetic
#pragma unroll
        for (int u = 0; u < FC_UNROLL; u++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (arp) av[u][j] = ((p.act == 3 && !(yv[u][j] > 0.f)) ? av[u][j] * p.alpha : av[u][j]) * p.gain;
                a_sum += av[u][j];
                a_sq = __builtin_fmaf(av[u][j], av[u][j], a_sq);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][0], bv[u][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][1], bv[u][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][2], bv[u][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][3], bv[u][3], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) acc0[e] += acc1[e];
    __syncthreads();   // rowstat zeroed
    // row statistics of A (sum / sum of squares over k): lanes r and r + 32 of all waves hold the pieces of row m0 + r
    if (p.normalize || p.colsum) {
        float v = p.normalize ? a_sq : a_sum;
        v += __shfl_xor(v, 32, 64);
        if (lane < 32) atomicAdd(&rowstat[lane], v);
    }
    // C layout of the 32x32 MFMA: col (n) = lane & 31, row (m) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).  Every wave parks its partial tile in LDS; then ALL
    // threads share the cross-wave sum and the epilogue, one output element per thread and pass.  With the sum on wave 0 alone (16 elements x 15 LDS reads per
    // lane) the tail of a K = 512 launch outlasted its k loop: 19.4 -> 13.6 us per launch over the 118 dense launches of a training iteration, 2.28 -> 1.59 ms
    // (profiles/r04_c19_fc_epilogue_ab.json).
    // (waves - 1 slots -- 60 KiB at sixteen waves, inside the default dynamic-LDS limit -- : waves 1.. park theirs, wave 0 adds its own into slot 0 behind the
    // first barrier; a single-wave launch keeps everything in registers' place: slot 0 is then its only tile)
    const int slots = waves > 1 ? waves - 1 : 1;
    if (wave > 0 || waves == 1) {
#pragma unroll
        for (int e = 0; e < 16; e++) red[(waves == 1 ? 0 : wave - 1) * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * kk) * 32 + r] = acc0[e];
    }
    __syncthreads();
    if (wave == 0 && waves > 1) {
#pragma unroll
        for (int e = 0; e < 16; e++) red[((e & 3) + 8 * (e >> 2) + 4 * kk) * 32 + r] += acc0[e];
    }
    if (waves > 1) __syncthreads();
    if (p.colsum && blockIdx.x == 0 && t < 32 && m0 + t < p.m) p.colsum[m0 + t] = rowstat[t] * p.bgain;
    for (int idx = t; idx < 1024; idx += (int)blockDim.x) {
        const int ml = idx >> 5, rr = idx & 31;
        const int cn = n0 + rr;
        float v = red[idx];
        for (int w = 1; w < slots; w++) v += red[w * 1024 + idx];
        if (p.normalize) v *= 1.0f / sqrtf(rowstat[ml] / (float)p.k + 1e-8f);
        const float bias = (p.bias && cn < p.n) ? p.bias[cn] * p.bgain : 0.f;
        v = v * p.wgain + bias;
        if (p.epilogue_act) v = ((p.act == 3 && !(v > 0.f)) ? v * p.alpha : v) * p.gain;
        if (m0 + ml < p.m && cn < p.n) {
            float* cq = p.c + bz * p.scb + (size_t)(m0 + ml) * p.scm + (size_t)cn * p.scn;
            *cq = p.accumulate ? *cq + v : v;
        }
    }
}

template <int AK, int BK, int FC_UNROLL>
__global__ __launch_bounds__(FC_WAVES * 64) void fc_kernel(fc_params p) { fc_body<AK, BK, FC_UNROLL>(p, blockIdx.z); }

// Several independent small products in ONE launch (round 6): the style affines of a synthesis pass -- 21 dense layers [frames, 512] x [512, C_l] with their own
// weight, bias, output and (ToRGB) output gain, each 10-14 us as a launch of its own whatever the batch: 107-118 dense launches per training iteration were 1.6 ms.
// blockIdx.z picks the problem; the table rides in the kernel arguments (capture-safe); the grid spans the largest problem, workgroups outside their own problem leave.
struct fc_group_item {
    const float* a; const float* aref; const float* b; float* c; const float* bias; float* colsum;
    int64_t sam, sak, sbk, sbn, scm;
    int m, n, k;
    float wgain, gain;
};
constexpr int FC_GROUP_MAX = 24;
struct fc_group {
    fc_params base;       // everything the problems share (activation, flags, bias gain, column strides)
    fc_group_item item[FC_GROUP_MAX];
};

template <int AK, int BK>
__global__ __launch_bounds__(FC_WAVES * 64) void fc_group_kernel(fc_group g) {
    const fc_group_item& it = g.item[blockIdx.z];
    if ((int)blockIdx.x * 32 >= it.n || (int)blockIdx.y * 32 >= it.m) return;      // (workgroup-uniform, in front of every barrier)
    fc_params p = g.base;
    p.a = it.a; p.aref = it.aref; p.b = it.b; p.c = it.c; p.bias = it.bias; p.colsum = it.colsum;
    p.sam = it.sam; p.sak = it.sak; p.sbk = it.sbk; p.sbn = it.sbn; p.scm = it.scm;
    p.m = it.m; p.n = it.n; p.k = it.k; p.wgain = it.wgain; p.gain = it.gain;
    fc_body<AK, BK, 1>(p, 0);
}

}  // namespace sgv_fck

static int sgv_fc_launch(const sgv_fc_params* q, hipStream_t stream, bool account) {
    const int batch = q->batch > 0 ? q->batch : 1;
    sgv_fck::fc_params p{};
    p.a = q->a; p.sam = q->a_stride_m; p.sak = q->a_stride_k; p.aref = q->a_ref;
    p.b = q->b; p.sbk = q->b_stride_k; p.sbn = q->b_stride_n;
    p.c = q->c; p.scm = q->c_stride_m; p.scn = q->c_stride_n;
    p.bias = q->bias; p.colsum = q->a_rowsum;
    p.m = q->m; p.n = q->n; p.k = q->k;
    p.normalize = q->normalize_a; p.act = q->act; p.alpha = q->alpha; p.gain = q->gain; p.wgain = q->weight_gain; p.bgain = q->bias_gain;
    p.epilogue_act = q->epilogue_act;
    p.sab = q->a_stride_batch; p.sbb = q->b_stride_batch; p.scb = q->c_stride_batch; p.accumulate = q->accumulate;
    // SGV_FC_UNROLL=4: four k steps' loads in flight per wave -- measured slower (tools/fc_bench.py: the [2048, 5632] x [5632, 512] products of the
    // motion network's conv1d layers 359 -> 631 us in the data-gradient form), kept for the lab only
    static const int unroll_env = [] { const char* e = getenv("SGV_FC_UNROLL"); return e ? atoi(e) : 1; }();
    static const int waves_env = [] { const char* e = getenv("SGV_FC_WAVES"); return e ? atoi(e) : 0; }();     // 0: follow K
    const int steps = (q->k + 7) / 8;
    // (all k steps of a wave in flight at once for K <= 512 -- SGV_FC_UNROLL=4 -- measured no better with either epilogue: 1.72 vs 1.60 ms of dense layers per
    // iteration, profiles/r04_c19_fc_epilogue_ab.json)
    const int unroll = unroll_env == 1 ? 1 : 4;
    const int waves = waves_env > 0 ? std::min(waves_env, sgv_fck::FC_WAVES) : std::max(1, std::min(sgv_fck::FC_WAVES, (steps + unroll - 1) / unroll));
    const size_t lds = (size_t)std::max(waves - 1, 1) * 1024 * sizeof(float);
    dim3 grid((unsigned)((q->n + 31) / 32), (unsigned)((q->m + 31) / 32), (unsigned)batch), block(waves * 64);
    const bool ak = q->a_stride_k == 1, bk = q->b_stride_k == 1;
    auto go = [&] {
        if (unroll == 1) {
            if (ak && bk) hipLaunchKernelGGL((sgv_fck::fc_kernel<1, 1, 1>), grid, block, lds, stream, p);
            else if (ak) hipLaunchKernelGGL((sgv_fck::fc_kernel<1, 0, 1>), grid, block, lds, stream, p);
            else if (bk) hipLaunchKernelGGL((sgv_fck::fc_kernel<0, 1, 1>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((sgv_fck::fc_kernel<0, 0, 1>), grid, block, lds, stream, p);
        } else {
            if (ak && bk) hipLaunchKernelGGL((sgv_fck::fc_kernel<1, 1, 4>), grid, block, lds, stream, p);
            else if (ak) hipLaunchKernelGGL((sgv_fck::fc_kernel<1, 0, 4>), grid, block, lds, stream, p);
            else if (bk) hipLaunchKernelGGL((sgv_fck::fc_kernel<0, 1, 4>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((sgv_fck::fc_kernel<0, 0, 4>), grid, block, lds, stream, p);
        }
    };
    if (account) {
        sgv_launch_scope scope(SGV_K_FC, stream, 4.0 * batch * ((double)q->m * q->k + (double)q->n * q->k + (double)q->m * q->n), 2.0 * batch * q->m * (double)q->n * q->k);
        go();
    } else {
        go();
    }
    sgv_note_variant(SGV_V_fc);
    return sgv_check_launch("fc_kernel");
}

extern "C" int sgv_fc(const sgv_fc_params* q, void* stream_) {
    if (!q) return sgv_fail(SGV_ERR_INVALID_ARG, "fc: params is NULL");
    if (!q->a || !q->b || !q->c) return sgv_fail(SGV_ERR_INVALID_ARG, "fc: NULL pointer");
    if (q->m < 1 || q->n < 1 || q->k < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "fc: sizes must be positive");
    if (q->act != 1 && q->act != 3) return sgv_fail(SGV_ERR_INVALID_ARG, "fc: act must be 1 (linear) or 3 (lrelu)");
    if ((q->m + 31) / 32 > 65535 || q->batch > 65535 || q->batch < 0) return sgv_fail(SGV_ERR_TOO_LARGE, "fc: too many rows / batches");
    if (q->batch > 1 && (q->a_rowsum || q->normalize_a)) return sgv_fail(SGV_ERR_UNSUPPORTED, "fc: row statistics are not batched");
    return sgv_fc_launch(q, (hipStream_t)stream_, true);
}

extern "C" int sgv_fc_grouped(const sgv_fc_params* q, int32_t count, void* stream_) {
    if (!q || count < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "fc_grouped: no problems");
    hipStream_t stream = (hipStream_t)stream_;
    const bool ak = q[0].a_stride_k == 1, bk = q[0].b_stride_k == 1;
    for (int i = 0; i < count; i++) {
        const sgv_fc_params& e = q[i];
        if (!e.a || !e.b || !e.c) return sgv_fail(SGV_ERR_INVALID_ARG, "fc_grouped: NULL pointer in problem %d", i);
        if (e.m < 1 || e.n < 1 || e.k < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "fc_grouped: sizes must be positive (problem %d)", i);
        if (e.batch > 1 || e.normalize_a) return sgv_fail(SGV_ERR_UNSUPPORTED, "fc_grouped: batched / normalising problems take sgv_fc");
        if ((e.a_stride_k == 1) != ak || (e.b_stride_k == 1) != bk || e.act != q[0].act || e.epilogue_act != q[0].epilogue_act || e.alpha != q[0].alpha
            || e.bias_gain != q[0].bias_gain || e.c_stride_n != q[0].c_stride_n || (e.a_ref == nullptr) != (q[0].a_ref == nullptr) || e.accumulate != q[0].accumulate)
            return sgv_fail(SGV_ERR_UNSUPPORTED, "fc_grouped: the problems of one call share the operand contiguity, the activation and the bias gain (problem %d differs)", i);
        if (e.act != 1 && e.act != 3) return sgv_fail(SGV_ERR_INVALID_ARG, "fc_grouped: act must be 1 (linear) or 3 (lrelu)");
    }
    for (int first = 0; first < count; first += sgv_fck::FC_GROUP_MAX) {
        const int cnt = std::min(count - first, sgv_fck::FC_GROUP_MAX);
        sgv_fck::fc_group g{};
        const sgv_fc_params& h = q[first];
        g.base.act = h.act; g.base.alpha = h.alpha; g.base.bgain = h.bias_gain; g.base.epilogue_act = h.epilogue_act; g.base.scn = h.c_stride_n; g.base.accumulate = h.accumulate;
        int max_m = 0, max_n = 0, max_k = 0;
        double bytes = 0.0, flops = 0.0;
        for (int i = 0; i < cnt; i++) {
            const sgv_fc_params& e = q[first + i];
            sgv_fck::fc_group_item& it = g.item[i];
            it.a = e.a; it.aref = e.a_ref; it.b = e.b; it.c = e.c; it.bias = e.bias; it.colsum = e.a_rowsum;
            it.sam = e.a_stride_m; it.sak = e.a_stride_k; it.sbk = e.b_stride_k; it.sbn = e.b_stride_n; it.scm = e.c_stride_m;
            it.m = e.m; it.n = e.n; it.k = e.k; it.wgain = e.weight_gain; it.gain = e.gain;
            max_m = std::max(max_m, e.m); max_n = std::max(max_n, e.n); max_k = std::max(max_k, e.k);
            bytes += 4.0 * ((double)e.m * e.k + (double)e.n * e.k + (double)e.m * e.n);
            flops += 2.0 * e.m * (double)e.n * e.k;
        }
        if ((max_m + 31) / 32 > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "fc_grouped: too many rows");
        const int steps = (max_k + 7) / 8;
        const int waves = std::max(1, std::min(sgv_fck::FC_WAVES, steps));
        const size_t lds = (size_t)std::max(waves - 1, 1) * 1024 * sizeof(float);
        dim3 grid((unsigned)((max_n + 31) / 32), (unsigned)((max_m + 31) / 32), (unsigned)cnt), block(waves * 64);
        sgv_launch_scope scope(SGV_K_FC, stream, bytes, flops);
        if (ak && bk) hipLaunchKernelGGL((sgv_fck::fc_group_kernel<1, 1>), grid, block, lds, stream, g);
        else if (ak) hipLaunchKernelGGL((sgv_fck::fc_group_kernel<1, 0>), grid, block, lds, stream, g);
        else if (bk) hipLaunchKernelGGL((sgv_fck::fc_group_kernel<0, 1>), grid, block, lds, stream, g);
        else hipLaunchKernelGGL((sgv_fck::fc_group_kernel<0, 0>), grid, block, lds, stream, g);
        sgv_note_variant(SGV_V_fc_grouped);
        const int rc = sgv_check_launch("fc_group_kernel");
        if (rc != SGV_OK) return rc;
    }
    return SGV_OK;
}
