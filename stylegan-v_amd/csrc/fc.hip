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
// Tiling: a workgroup owns a 32 x 32 output tile (grid = N/32 x M/32: 16..256 workgroups for the layers above); its four waves split K
// four ways, each accumulates a whole 32x32 MFMA tile (16 registers), partial tiles are added through LDS, then the epilogue.
// Algorithmic bytes: 4 * (M*K + N*K + M*N); flops 2*M*N*K.

#include "sgv_common.h"

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
};

__global__ __launch_bounds__(256) void fc_kernel(fc_params p) {
    __shared__ float red[3][32 * 32 + 32];
    __shared__ float rowstat[32];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, kk = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int am = m0 + r, bn = n0 + r;
    const bool a_ok = am < p.m, b_ok = bn < p.n;
    const float* ap = p.a + (size_t)(a_ok ? am : 0) * p.sam;
    const float* arp = p.aref ? p.aref + (size_t)(a_ok ? am : 0) * p.sam : nullptr;
    const float* bp = p.b + (size_t)(b_ok ? bn : 0) * p.sbn;

    // wave w takes k = 2 * (4 * i + w) + kk: interleaved so that the four waves stream neighbouring addresses
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    float a_sum = 0.f, a_sq = 0.f;
    const int steps = (p.k + 7) / 8;
    for (int i = 0; i < steps; i++) {
        const int k = 2 * (4 * i + wave) + kk;
        const bool k_ok = k < p.k;
        float av = (a_ok && k_ok) ? ap[(size_t)k * p.sak] : 0.f;
        if (arp) {
            const float yv = (a_ok && k_ok) ? arp[(size_t)k * p.sak] : 0.f;
            av = ((p.act == 3 && !(yv > 0.f)) ? av * p.alpha : av) * p.gain;
        }
        const float bv = (b_ok && k_ok) ? bp[(size_t)k * p.sbk] : 0.f;
        a_sum += av;
        a_sq = __builtin_fmaf(av, av, a_sq);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    // row statistics of A (sum / sum of squares over k): lanes r and r + 32 of the four waves hold the pieces of row m0 + r
    if (p.normalize || p.colsum) {
        float v = p.normalize ? a_sq : a_sum;
        v += __shfl_xor(v, 32, 64);
        if (wave == 0 && lane < 32) rowstat[lane] = 0.f;
        __syncthreads();
        if (lane < 32) atomicAdd(&rowstat[lane], v);
    }
    // C layout of the 32x32 MFMA: col (n) = lane & 31, row (m) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) red[wave - 1][((e & 3) + 8 * (e >> 2) + 4 * kk) * 32 + r] = acc[e];
    }
    __syncthreads();
    if (wave != 0) return;
    if (p.colsum && blockIdx.x == 0 && lane < 32 && m0 + lane < p.m) p.colsum[m0 + lane] = rowstat[lane] * p.bgain;
    const float bias = (p.bias && b_ok) ? p.bias[bn] * p.bgain : 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int ml = (e & 3) + 8 * (e >> 2) + 4 * kk;
        float v = acc[e] + ((red[0][ml * 32 + r] + red[1][ml * 32 + r]) + red[2][ml * 32 + r]);
        if (p.normalize) v *= 1.0f / sqrtf(rowstat[ml] / (float)p.k + 1e-8f);
        v = v * p.wgain + bias;
        if (p.epilogue_act) v = ((p.act == 3 && !(v > 0.f)) ? v * p.alpha : v) * p.gain;
        if (m0 + ml < p.m && b_ok) p.c[(size_t)(m0 + ml) * p.scm + (size_t)bn * p.scn] = v;
    }
}

}  // namespace sgv_fck

extern "C" int sgv_fc(const sgv_fc_params* q, void* stream_) {
    if (!q) return sgv_fail(SGV_ERR_INVALID_ARG, "fc: params is NULL");
    if (!q->a || !q->b || !q->c) return sgv_fail(SGV_ERR_INVALID_ARG, "fc: NULL pointer");
    if (q->m < 1 || q->n < 1 || q->k < 1) return sgv_fail(SGV_ERR_INVALID_ARG, "fc: sizes must be positive");
    if (q->act != 1 && q->act != 3) return sgv_fail(SGV_ERR_INVALID_ARG, "fc: act must be 1 (linear) or 3 (lrelu)");
    if ((q->m + 31) / 32 > 65535) return sgv_fail(SGV_ERR_TOO_LARGE, "fc: too many rows");
    sgv_fck::fc_params p{};
    p.a = q->a; p.sam = q->a_stride_m; p.sak = q->a_stride_k; p.aref = q->a_ref;
    p.b = q->b; p.sbk = q->b_stride_k; p.sbn = q->b_stride_n;
    p.c = q->c; p.scm = q->c_stride_m; p.scn = q->c_stride_n;
    p.bias = q->bias; p.colsum = q->a_rowsum;
    p.m = q->m; p.n = q->n; p.k = q->k;
    p.normalize = q->normalize_a; p.act = q->act; p.alpha = q->alpha; p.gain = q->gain; p.wgain = q->weight_gain; p.bgain = q->bias_gain;
    p.epilogue_act = q->epilogue_act;
    hipStream_t stream = (hipStream_t)stream_;
    sgv_launch_scope scope(SGV_K_GEMM, stream, 4.0 * ((double)q->m * q->k + (double)q->n * q->k + (double)q->m * q->n), 2.0 * q->m * (double)q->n * q->k);
    dim3 grid((unsigned)((q->n + 31) / 32), (unsigned)((q->m + 31) / 32));
    hipLaunchKernelGGL(sgv_fck::fc_kernel, grid, dim3(256), 0, stream, p);
    return sgv_check_launch("fc_kernel");
}
