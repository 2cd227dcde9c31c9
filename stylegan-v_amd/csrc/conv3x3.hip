// C ABI entry of the 3x3 / stride-1 / pad-1 convolution and its data gradient (kernel: conv3x3_kernel.h).
//
// Replaces, for the shapes it supports, `torch.nn.functional.conv2d` / `conv_transpose2d` as called by the reference's
// conv2d_gradfix (src/torch_utils/ops/conv2d_gradfix.py:35-43, 100-118) for the 3x3 layers of networks.py / layers.py.

#include "sgv_common.h"
#include "conv3x3_kernel.h"
#include "conv3x3_ws_kernel.h"
#include "conv3x3s2_kernel.h"
#include "conv3x3s2_ws_kernel.h"

#include <algorithm>
#include <mutex>
#include <stdlib.h>

using namespace sgv_conv;

namespace {

// small square images (16x16, 8x8, 4x4): conv3x3_small_kernel packs 2 / 8 / 32 whole samples into a tile; a last tile may be partly filled
int small_samples(int n, int h, int w) {
    if (h == 16 && w == 16 && n >= 1) return small_cfg<16>::S;
    if (h == 8 && w == 8 && n >= 1) return small_cfg<8>::S;
    if (h == 4 && w == 4 && n >= 1) return small_cfg<4>::S;
    return 0;
}

bool io16(int dtype) { return dtype == SGV_BF16 || dtype == SGV_F16; }

// 16-bit tensors (bf16 / fp16 activations, fp32 weights): the producer / consumer kernels only (images >= 32 pixels), single 16-bit operands (terms = 1: bf16 tensors as
// bf16, fp16 tensors as fp16 on the f16 MFMA -- operand_format of sgv_split.h)
// Output channels: whole 64-channel tiles; the producer / consumer kernels (images >= 32 pixels) also take a half-full last tile (m % 32 == 0: the 32-channel
// layers of the 1024^2 synthesis network, BASELINE configs[4]) -- its upper 32 rows are zero weights and are not stored.  SGV_CONV_M32=0 switches that off.
int tiles_m(int m) { return (m + TM - 1) / TM; }
bool half_tiles_on() { static const bool on = !(getenv("SGV_CONV_M32") && getenv("SGV_CONV_M32")[0] == '0'); return on; }

bool supported(int n, int k, int m, int h, int w, int dtype) {
    if (!((dtype == SGV_F32 || io16(dtype)) && n >= 1 && k >= KC && k % KC == 0 && m >= 32 && m % 32 == 0 && (int64_t)n * std::max(k, m) * h * w <= INT32_MAX)) return false;
    const bool whole = m % TM == 0;
    if (w >= SEG && w % SEG == 0 && h >= TROWS && h % TROWS == 0) return whole || half_tiles_on();
    return whole && dtype == SGV_F32 && small_samples(n, h, w) > 0;
}

std::once_flag g_attr_once;
int g_cus = 256;
hipError_t g_attr_err = hipSuccess;
bool g_edge_mfma = true;   // SGV_CONVT_EDGE_MFMA=0: the gather + FMA edge kernels instead of convT3x3_s2_edge_mfma
bool g_s2_ws = true;    // SGV_S2_WS=0: the one-role-per-wave stride-2 kernels of conv3x3s2_kernel.h
bool g_use_ws = true;   // SGV_CONV_WS=0: the 4-wave kernel of conv3x3_kernel.h instead of the producer / consumer form

bool big_image(int h, int w) { return w >= SEG && w % SEG == 0 && h >= TROWS && h % TROWS == 0; }

typedef void (*ws_kernel_t)(conv_ws_params);
// [terms: 1 | 3 | 4][PRO][EPI]
const ws_kernel_t g_ws_kernels[3][2][2] = {
    {{conv3x3_ws_kernel<1, 0, 0>, conv3x3_ws_kernel<1, 0, 1>}, {conv3x3_ws_kernel<1, 1, 0>, conv3x3_ws_kernel<1, 1, 1>}},
    {{conv3x3_ws_kernel<3, 0, 0>, conv3x3_ws_kernel<3, 0, 1>}, {conv3x3_ws_kernel<3, 1, 0>, conv3x3_ws_kernel<3, 1, 1>}},
    {{conv3x3_ws_kernel<4, 0, 0>, conv3x3_ws_kernel<4, 0, 1>}, {conv3x3_ws_kernel<4, 1, 0>, conv3x3_ws_kernel<4, 1, 1>}}};
int terms_index(int terms) { return terms == 1 ? 0 : terms == 3 ? 1 : 2; }
// 16-bit tensors: [bf16 | fp16][PRO][EPI]
const ws_kernel_t g_ws_kernels_io[2][2][2] = {
    {{conv3x3_ws_kernel<1, 0, 0, 0, 1, 1>, conv3x3_ws_kernel<1, 0, 1, 0, 1, 1>}, {conv3x3_ws_kernel<1, 1, 0, 0, 1, 1>, conv3x3_ws_kernel<1, 1, 1, 0, 1, 1>}},
    {{conv3x3_ws_kernel<1, 0, 0, 0, 1, 2>, conv3x3_ws_kernel<1, 0, 1, 0, 1, 2>}, {conv3x3_ws_kernel<1, 1, 0, 0, 1, 2>, conv3x3_ws_kernel<1, 1, 1, 0, 1, 2>}}};

void init_once() {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<16>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<3, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<16>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<8>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<8>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<16>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<8>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_small_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, small_cfg<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
    for (int t = 0; t < 3; t++) for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) {
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)g_ws_kernels[t][a][b], hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
        if (e == hipSuccess && t < 2) e = hipFuncSetAttribute((const void*)g_ws_kernels_io[t][a][b], hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
    }
    g_attr_err = e;
    const char* env = getenv("SGV_CONV_WS");
    g_use_ws = !(env && env[0] == '0');
    env = getenv("SGV_S2_WS");
    g_s2_ws = !(env && env[0] == '0');
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_ws_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, S2W_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_ws_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, S2W_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_ws_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, S2W_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<4, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(2));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<4, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(4));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<4, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<1, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(2));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<3, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(2));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<1, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(4));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<3, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(4));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<3, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
#define SGV_P2_ATTR(T, E, SS) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<T, 0, E, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, p2_lds_bytes(SS));
    SGV_P2_ATTR(1, 0, 2) SGV_P2_ATTR(3, 0, 2) SGV_P2_ATTR(1, 1, 2) SGV_P2_ATTR(3, 1, 2) SGV_P2_ATTR(1, 0, 4) SGV_P2_ATTR(3, 0, 4) SGV_P2_ATTR(1, 1, 4) SGV_P2_ATTR(3, 1, 4)
    SGV_P2_ATTR(4, 0, 2) SGV_P2_ATTR(4, 1, 2) SGV_P2_ATTR(4, 0, 4) SGV_P2_ATTR(4, 1, 4)
#undef SGV_P2_ATTR
#define SGV_P2_ATTR_IO(E, SS) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<1, 0, E, SS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, p2_lds_bytes(SS)); \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_pairs_kernel<1, 0, E, SS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, p2_lds_bytes(SS));
    SGV_P2_ATTR_IO(0, 1) SGV_P2_ATTR_IO(1, 1) SGV_P2_ATTR_IO(0, 2) SGV_P2_ATTR_IO(1, 2) SGV_P2_ATTR_IO(0, 4) SGV_P2_ATTR_IO(1, 4)
#undef SGV_P2_ATTR_IO
#define SGV_TW_ATTR_IO(SS) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<1, 0, SS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(SS)); \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)convT3x3_s2_ws_kernel<1, 0, SS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, tw_lds_bytes(SS));
    SGV_TW_ATTR_IO(1) SGV_TW_ATTR_IO(2) SGV_TW_ATTR_IO(4)
#undef SGV_TW_ATTR_IO
    g_attr_err = e;
    env = getenv("SGV_CONVT_EDGE_MFMA");
    g_edge_mfma = !(env && env[0] == '0');
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) g_cus = prop.multiProcessorCount;
}

bool supported_s2(int n, int k, int m, int h, int w, int dtype, int mode = 0) {   // h, w: the small (H x W) grid; mode 2 (transposed, producer / consumer kernel): also m % 32 == 0
    const bool m_ok = m >= TM && m % TM == 0 ? true : (mode == 2 && g_s2_ws && g_edge_mfma && half_tiles_on() && m >= 32 && m % 32 == 0);
    return (dtype == SGV_F32 || io16(dtype)) && n >= 1 && k >= KC && k % KC == 0 && m_ok && w >= SEG && w % SEG == 0 && h >= S_ROWS && h % S_ROWS == 0 &&
           (int64_t)n * std::max(k, m) * (2 * h + 1) * (2 * w + 1) <= INT32_MAX;
}

// W = 16 / 8: the producer / consumer kernels pack 2 / 4 samples into a 32-pixel tile row (transposed form so far)
bool supported_s2_packed(int n, int k, int m, int h, int w, int mode, int dtype) {
    if (!((dtype == SGV_F32 || io16(dtype)) && (w == 16 || w == 8) && n >= 1 && n % (32 / w) == 0 && k >= KC && k % KC == 0)) return false;
    if (mode == 2) return m >= TM && m % TM == 0 && h >= TW_ROWS && h % TW_ROWS == 0;
    return mode == 0 && m % P2_TM == 0 && h % P2_ROWS == 0;   // the strided form: the tap-pair kernel only
}

bool pairs_shape(int c_out, int h) { return g_s2_ws && c_out % P2_TM == 0 && h % P2_ROWS == 0; }

// 16-bit tensors: the producer / consumer kernels only -- the tap-pair kernel for the strided form (c_out % 128, H % 8), convT3x3_s2_ws_kernel + the MFMA
// edge strips for the transposed one
bool s2_mode_ok(int c_out, int h, int mode, int dtype) { return !io16(dtype) || (g_s2_ws && (mode == 2 ? g_edge_mfma : pairs_shape(c_out, h))); }

}  // namespace

// max |x| of a dense tensor as one fp32 in device memory: the bound the block-scaled fp16 split (terms = 4, sgv_split.h) scales an operand by
extern "C" int sgv_absmax(const void* x, int64_t numel, int dtype, float* out, int32_t accumulate, void* stream_) {
    if (!out || (!x && numel > 0) || numel < 0) return sgv_fail(SGV_ERR_INVALID_ARG, "absmax: NULL pointer or negative size");
    if (dtype != SGV_F32 && dtype != SGV_F16 && dtype != SGV_BF16) return sgv_fail(SGV_ERR_UNSUPPORTED, "absmax: fp32 / fp16 / bf16 tensors only");
    hipStream_t stream = (hipStream_t)stream_;
    if (!accumulate && hipMemsetAsync(out, 0, 4, stream) != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "absmax: hipMemsetAsync failed");
    if (numel == 0) return SGV_OK;
    const int64_t epv = dtype == SGV_F32 ? 4 : 8;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((numel / epv + 1023) / 1024, 1024));     // four 16-byte vectors per lane and pass; <= 1024 workgroups (four per CU)
    sgv_launch_scope scope(SGV_K_ABSMAX, stream, (double)numel * (dtype == SGV_F32 ? 4.0 : 2.0), 0.0, false);
    if (dtype == SGV_F32) hipLaunchKernelGGL(absmax_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, (size_t)numel, (unsigned*)out);
    else if (dtype == SGV_F16) hipLaunchKernelGGL(absmax_kernel<_Float16>, dim3(blocks), dim3(256), 0, stream, (const _Float16*)x, (size_t)numel, (unsigned*)out);
    else hipLaunchKernelGGL(absmax_kernel<__bf16>, dim3(blocks), dim3(256), 0, stream, (const __bf16*)x, (size_t)numel, (unsigned*)out);
    return sgv_check_launch("absmax_kernel");
}

extern "C" int sgv_conv3x3_supported(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int dtype) {
    return supported(n, c_in, c_out, h, w, dtype) ? 1 : 0;
}

// 16-bit hi + lo per weight, whole 64-row tiles; the last 16 bytes hold the weight's bound for the block-scaled split (terms = 4)
constexpr int64_t AMAX_TAIL = 16;
extern "C" int64_t sgv_conv3x3_workspace_bytes(int32_t c_in, int32_t c_out) {
    return (int64_t)c_in * ((c_out + TM - 1) / TM * TM) * 9 * 4 + AMAX_TAIL;
}

namespace {
// terms = 4: max |w| into the workspace tail (cleared first), for the weight preparation and the kernel's epilogue
int weight_amax(const float* w, size_t numel, float* slot, hipStream_t stream) {
    if (hipMemsetAsync(slot, 0, 4, stream) != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "conv3x3: hipMemsetAsync failed");
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((numel / 4 + 1023) / 1024, 256));
    hipLaunchKernelGGL(absmax_kernel<float>, dim3(blocks), dim3(256), 0, stream, w, numel, (unsigned*)slot);
    return sgv_check_launch("absmax_kernel (weights)");
}
}  // namespace

namespace {

int conv3x3_impl(const sgv_conv3x3_params* p, const sgv_conv3x3_epilogue* ep, int dtype, void* stream_, const char* who) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: params is NULL", who);
    if (!p->x || !p->weight || !p->y || !p->workspace) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: NULL pointer", who);
    if (!supported(p->n, p->c_in, p->c_out, p->h, p->w, dtype))
        return sgv_fail(SGV_ERR_UNSUPPORTED, "%s: needs fp32 / bf16 / fp16 tensors, c_in %% 16 == 0, c_out %% 32 == 0, and W %% 32 == 0, H %% 16 == 0 (fp32 with c_out %% 64 == 0 also: 16x16 / 8x8 / 4x4 images) (got n=%d c_in=%d c_out=%d h=%d w=%d dtype=%d)",
                        who, p->n, p->c_in, p->c_out, p->h, p->w, dtype);
    if (io16(dtype) && p->terms != 1) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: 16-bit tensors need terms = 1 (one 16-bit operand per value: bf16, or fp16 for fp16 tensors)", who);
    if (p->terms == 4 && !p->x_amax) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: terms = 4 (block-scaled fp16 split) needs x_amax, a device pointer to an upper bound of max |x| (sgv_absmax)", who);
    if (io16(dtype) && ep && ep->accumulate) return sgv_fail(SGV_ERR_UNSUPPORTED, "%s: accumulate needs fp32 tensors", who);
    if (ep && !big_image(p->h, p->w)) return sgv_fail(SGV_ERR_UNSUPPORTED, "%s: the fused form needs W %% 32 == 0 and H %% 16 == 0 (got h=%d w=%d)", who, p->h, p->w);
    if (ep && (ep->act != 1 && ep->act != 3)) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: act must be 1 (linear) or 3 (lrelu)", who);
    if (ep && (!(ep->gain > 0.f) || (ep->act == 3 && !(ep->alpha >= 0.f && ep->alpha <= 1.f)))) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: needs gain > 0 and 0 <= alpha <= 1", who);
    if (p->mode != 0 && p->mode != 1) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: mode must be 0 (forward) or 1 (data gradient)", who);
    if (p->terms != 1 && p->terms != 3 && p->terms != 4) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: terms must be 1, 3 or 4", who);
    if (p->workspace_bytes < sgv_conv3x3_workspace_bytes(p->c_in, p->c_out)) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: workspace is too small", who);
    if ((((uintptr_t)p->x) | ((uintptr_t)p->y) | ((uintptr_t)p->workspace)) & 15) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: x, y and workspace must be 16-byte aligned", who);
    if (ep && ep->x_scale && (((uintptr_t)ep->x_scale) & 15)) return sgv_fail(SGV_ERR_INVALID_ARG, "%s: x_scale must be 16-byte aligned", who);
    std::call_once(g_attr_once, init_once);
    if (g_attr_err != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "%s: hipFuncSetAttribute failed: %s", who, hipGetErrorString(g_attr_err));
    hipStream_t stream = (hipStream_t)stream_;

    const int words = tiles_m(p->c_out) * (p->c_in / KC) * 9 * 2 * TM;
    const float* w_amax = p->w_amax;          // the caller's bound of the weight, or our own pass into the workspace tail
    int rc = SGV_OK;
    if (p->terms == 4 && !w_amax) {
        float* slot = (float*)((char*)p->workspace + sgv_conv3x3_workspace_bytes(p->c_in, p->c_out) - AMAX_TAIL);
        if ((rc = weight_amax(p->weight, (size_t)p->c_in * p->c_out * 9, slot, stream)) != SGV_OK) return rc;
        w_amax = slot;
    }
    hipLaunchKernelGGL(conv3x3_prep_weights, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, p->weight, (u32x4*)p->workspace, p->c_out, p->c_in, p->mode,
                       dtype == SGV_F16 ? 2 : p->terms, w_amax);       // fp16 tensors: fp16 operands (format 2 of sgv_split.h)
    rc = sgv_check_launch("conv3x3_prep_weights");
    if (rc != SGV_OK) return rc;

    conv_params kp{};
    kp.x = (const float*)p->x; kp.wprep = (const u32x4*)p->workspace; kp.y = (float*)p->y;
    kp.n = p->n; kp.k = p->c_in; kp.m = p->c_out; kp.h = p->h; kp.w = p->w;
    if (p->terms == 4 && ep && ep->x_scale && !p->x_amax2)
        return sgv_fail(SGV_ERR_INVALID_ARG, "%s: terms = 4 with x_scale needs x_amax2, a device pointer to an upper bound of max |x_scale| (the operand is x * x_scale)", who);
    if (p->terms == 4) { kp.x_amax = p->x_amax; kp.x_amax2 = p->x_amax2; kp.w_amax = w_amax; }
    const int small = big_image(p->h, p->w) ? 0 : small_samples(p->n, p->h, p->w);
    kp.tiles = small ? ((p->n + small - 1) / small) * (p->c_out / TM) : p->n * (p->h / TROWS) * (p->w / SEG) * tiles_m(p->c_out);
    if (small) {
        // fewer tiles than half the CUs: share every tile's input channels out over 2 / 4 / 8 workgroups (partial sums meet in a zeroed y through atomics).
        // SGV_CONV_SMALL_KSPLIT = 1 switches it off, 2 / 4 / 8 forces a split wherever the chunk count allows it.
        static const int forced = [] { const char* e = getenv("SGV_CONV_SMALL_KSPLIT"); return e ? atoi(e) : 0; }();
        const int chunks = p->c_in / KC;
        int ks = 1;
        if (forced > 1) { if (chunks % forced == 0) ks = forced; }
        else if (forced == 0) {
            for (int cand = 8; cand > 1; cand >>= 1)
                if (chunks % cand == 0 && chunks / cand >= 2 && (int64_t)kp.tiles * cand <= 2 * g_cus) { ks = cand; break; }
        }
        kp.ksplit = ks;
        kp.tiles *= ks;
    }
    kp.grid = std::min(kp.tiles, g_cus);
    const double elems = (double)p->n * p->h * p->w;
    const double es = io16(dtype) ? 2.0 : 4.0;
    sgv_launch_scope scope(small ? SGV_K_CONV3X3 : SGV_K_CONV3X3_S1, stream, es * elems * (p->c_in + p->c_out) + 4.0 * p->c_in * p->c_out * 9, 2.0 * elems * p->c_in * (double)p->c_out * 9,
                           true, /* own_stamps: conv3x3_ws_kernel writes its own timestamp pair when it is timed inside a capture */ !small);
    if (small) {
        if (kp.ksplit > 1 && hipMemsetAsync(p->y, 0, (size_t)p->n * p->c_out * p->h * p->w * sizeof(float), stream) != hipSuccess)
            return sgv_fail(SGV_ERR_LAUNCH, "conv3x3: clearing y for the split-K small-image kernel failed");
        if (p->w == 16) {
            if (p->terms == 1) hipLaunchKernelGGL((conv3x3_small_kernel<1, 16>), dim3((unsigned)kp.grid), dim3(256), small_cfg<16>::LDS, stream, kp);
            else if (p->terms == 3) hipLaunchKernelGGL((conv3x3_small_kernel<3, 16>), dim3((unsigned)kp.grid), dim3(256), small_cfg<16>::LDS, stream, kp);
            else hipLaunchKernelGGL((conv3x3_small_kernel<4, 16>), dim3((unsigned)kp.grid), dim3(256), small_cfg<16>::LDS, stream, kp);
        } else if (p->w == 8) {
            if (p->terms == 1) hipLaunchKernelGGL((conv3x3_small_kernel<1, 8>), dim3((unsigned)kp.grid), dim3(256), small_cfg<8>::LDS, stream, kp);
            else if (p->terms == 3) hipLaunchKernelGGL((conv3x3_small_kernel<3, 8>), dim3((unsigned)kp.grid), dim3(256), small_cfg<8>::LDS, stream, kp);
            else hipLaunchKernelGGL((conv3x3_small_kernel<4, 8>), dim3((unsigned)kp.grid), dim3(256), small_cfg<8>::LDS, stream, kp);
        } else {      // 4 x 4: the first generator block and the discriminator's epilogue (networks.py:518-576), 32 samples per tile
            if (p->terms == 1) hipLaunchKernelGGL((conv3x3_small_kernel<1, 4>), dim3((unsigned)kp.grid), dim3(256), small_cfg<4>::LDS, stream, kp);
            else if (p->terms == 3) hipLaunchKernelGGL((conv3x3_small_kernel<3, 4>), dim3((unsigned)kp.grid), dim3(256), small_cfg<4>::LDS, stream, kp);
            else hipLaunchKernelGGL((conv3x3_small_kernel<4, 4>), dim3((unsigned)kp.grid), dim3(256), small_cfg<4>::LDS, stream, kp);
        }
        sgv_note_variant(SGV_V_conv_small);
        return sgv_check_launch("conv3x3_small_kernel");
    }
    if (ep || g_use_ws || io16(dtype) || p->c_out % TM != 0 || p->terms == 4) {     // (half-full m tiles, the block-scaled split: the producer / consumer kernel only)
        conv_ws_params wp{};
        wp.c = kp;
        wp.stamp = scope.kernel_stamps();
        int pro = 0, epi = 0;
        if (ep) {
            wp.xscale = ep->x_scale; wp.oscale = ep->out_scale; wp.bias = ep->bias;
            wp.act = ep->act; wp.alpha = ep->alpha; wp.gain = ep->gain; wp.clamp = ep->clamp; wp.accumulate = ep->accumulate;
            pro = ep->x_scale ? 1 : 0;
            // a bare convolution (no scales, no bias, linear, gain 1, no clamp: the accumulate-into data gradient) keeps the plain store path
            epi = (ep->out_scale || ep->bias || ep->act != 1 || ep->gain != 1.f || ep->clamp >= 0.f) ? 1 : 0;
        }
        if (io16(dtype)) {
            hipLaunchKernelGGL(g_ws_kernels_io[dtype == SGV_F16][pro][epi], dim3((unsigned)kp.grid), dim3(512), WS_LDS_BYTES, stream, wp);
            sgv_note_variant(SGV_V_conv_lowp);
            if (pro || epi) sgv_note_variant(SGV_V_conv_s1_ws_fused);
            if (p->c_out % TM != 0) sgv_note_variant(SGV_V_conv_s1_half_tile);
            return sgv_check_launch("conv3x3_ws_kernel (16-bit tensors)");
        }
        wp.y_amax = scope.take_amax_sink();      // (fp32 tensors here)
        hipLaunchKernelGGL(g_ws_kernels[terms_index(p->terms)][pro][epi], dim3((unsigned)kp.grid), dim3(512), WS_LDS_BYTES, stream, wp);
        sgv_note_variant((pro || epi) ? SGV_V_conv_s1_ws_fused : (ep && ep->accumulate) ? SGV_V_conv_s1_ws_accumulate : SGV_V_conv_s1_ws);
        if (p->c_out % TM != 0) sgv_note_variant(SGV_V_conv_s1_half_tile);     // (the last 64-row tile is half full: the 32-channel layers at 1024^2)
        return sgv_check_launch("conv3x3_ws_kernel");
    }
    scope.begin_stamp();      // (the four-wave kernel has no timestamp output: bracketed like every other launch)
    if (p->terms == 1) hipLaunchKernelGGL(conv3x3_kernel<1>, dim3((unsigned)kp.grid), dim3(256), LDS_BYTES, stream, kp);
    else hipLaunchKernelGGL(conv3x3_kernel<3>, dim3((unsigned)kp.grid), dim3(256), LDS_BYTES, stream, kp);
    sgv_note_variant(SGV_V_conv_s1_4wave);
    return sgv_check_launch("conv3x3_kernel");
}

}  // namespace

extern "C" int sgv_conv3x3(const sgv_conv3x3_params* p, int dtype, void* stream_) { return conv3x3_impl(p, nullptr, dtype, stream_, "conv3x3"); }

extern "C" int sgv_conv3x3_fused(const sgv_conv3x3_params* p, const sgv_conv3x3_epilogue* e, int dtype, void* stream_) {
    if (!e) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_fused: epilogue is NULL");
    return conv3x3_impl(p, e, dtype, stream_, "conv3x3_fused");
}

extern "C" int sgv_conv3x3_fused_supported(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int dtype) {
    return supported(n, c_in, c_out, h, w, dtype) && big_image(h, w) ? 1 : 0;
}

extern "C" int sgv_conv3x3_s2_supported(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int dtype) {   // both forms
    std::call_once(g_attr_once, init_once);
    return supported_s2(n, c_in, c_out, h, w, dtype) && s2_mode_ok(c_out, h, 0, dtype) && s2_mode_ok(c_out, h, 2, dtype) ? 1 : 0;
}

extern "C" int64_t sgv_conv3x3_s2_workspace_bytes(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int32_t mode) {
    int64_t bytes = (int64_t)c_in * ((c_out + TM - 1) / TM * TM) * 10 * 4;   // nine taps, ten in the tap-pair layout of conv3x3_s2_pairs_kernel; whole 64-row tiles
    if (mode == 2) bytes += (int64_t)convT3x3_s2_edge_floats(n, c_in, c_out, h, w) * 4;
    return (bytes + 15) / 16 * 16 + AMAX_TAIL;       // + the weight's bound for the block-scaled split (terms = 4)
}

namespace {


int conv3x3_s2_impl(const sgv_conv3x3_params* p, const sgv_conv3x3_s2_epilogue* ep, int dtype, void* stream_) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2: params is NULL");
    if (!p->x || !p->weight || !p->y || !p->workspace) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2: NULL pointer");
    std::call_once(g_attr_once, init_once);
    if (p->mode != 0 && p->mode != 2) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2: mode must be 0 (strided convolution) or 2 (transposed convolution)");
    const bool packed = g_s2_ws && supported_s2_packed(p->n, p->c_in, p->c_out, p->h, p->w, p->mode, dtype);
    if ((!packed && !supported_s2(p->n, p->c_in, p->c_out, p->h, p->w, dtype, p->mode)) || !s2_mode_ok(p->c_out, p->h, p->mode, dtype))
        return sgv_fail(SGV_ERR_UNSUPPORTED, "conv3x3_s2: needs c_in %% 16 == 0, c_out %% 64 == 0 (transposed form: %% 32), W %% 32 == 0, H %% 8 == 0 on the HxW grid; 16-bit tensors: the strided form with c_out %% 128 == 0, the transposed form with the MFMA edge strips (got n=%d c_in=%d c_out=%d h=%d w=%d mode=%d dtype=%d)",
                        p->n, p->c_in, p->c_out, p->h, p->w, p->mode, dtype);
    if (p->terms != 1 && p->terms != 3 && p->terms != 4) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2: terms must be 1, 3 or 4");
    if (io16(dtype) && p->terms != 1) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2: 16-bit tensors need terms = 1 (one 16-bit operand per value: bf16, or fp16 for fp16 tensors)");
    if (p->terms == 4 && !p->x_amax) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2: terms = 4 (block-scaled fp16 split) needs x_amax, a device pointer to an upper bound of max |x| (sgv_absmax)");
    if (p->terms == 4 && (!g_s2_ws || (p->mode == 2 && !g_edge_mfma))) return sgv_fail(SGV_ERR_UNSUPPORTED, "conv3x3_s2: terms = 4 is served by the producer / consumer kernels only (SGV_S2_WS / SGV_CONVT_EDGE_MFMA are off)");
    if (io16(dtype) && ep && ep->accumulate) return sgv_fail(SGV_ERR_UNSUPPORTED, "conv3x3_s2_fused: accumulate needs fp32 tensors");
    const int io = dtype == SGV_BF16 ? 1 : dtype == SGV_F16 ? 2 : 0;
    if (p->workspace_bytes < sgv_conv3x3_s2_workspace_bytes(p->n, p->c_in, p->c_out, p->h, p->w, p->mode)) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2: workspace is too small");
    if ((((uintptr_t)p->workspace) & 15) || (p->mode == 2 && (((uintptr_t)p->x) & (io ? 7 : 15)))) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2: workspace (and x of the transposed form: four elements) must be 16-byte aligned");
    std::call_once(g_attr_once, init_once);
    if (g_attr_err != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "conv3x3_s2: hipFuncSetAttribute failed: %s", hipGetErrorString(g_attr_err));
    hipStream_t stream = (hipStream_t)stream_;

    const bool pairs = p->mode == 0 && pairs_shape(p->c_out, p->h) && (p->w >= SEG || packed);
    if (ep) {
        if (!pairs) return sgv_fail(SGV_ERR_UNSUPPORTED, "conv3x3_s2_fused: needs the strided form (mode 0), c_out %% 128 == 0 and H %% 8 == 0 (and SGV_S2_WS != 0)");
        if (ep->act != 1 && ep->act != 3) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2_fused: act must be 1 (linear) or 3 (lrelu)");
        if (!(ep->gain > 0.f) || (ep->act == 3 && !(ep->alpha >= 0.f && ep->alpha <= 1.f))) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2_fused: needs gain > 0 and 0 <= alpha <= 1");
        if (ep->bias && (((uintptr_t)ep->bias) & 15)) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2_fused: bias must be 16-byte aligned");
    }
    int rc = SGV_OK;
    const float* w_amax = p->w_amax;
    if (p->terms == 4 && !w_amax) {
        float* slot = (float*)((char*)p->workspace + sgv_conv3x3_s2_workspace_bytes(p->n, p->c_in, p->c_out, p->h, p->w, p->mode) - AMAX_TAIL);
        if ((rc = weight_amax(p->weight, (size_t)p->c_in * p->c_out * 9, slot, stream)) != SGV_OK) return rc;
        w_amax = slot;
    }
    if (pairs) {
        const int words = (p->c_out / P2_TM) * (p->c_in / P2_KC) * 10 * P2_TM;
        hipLaunchKernelGGL(conv3x3_prep_weights_pairs, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, p->weight, (u32x4*)p->workspace, p->c_out, p->c_in,
                           dtype == SGV_F16 ? 2 : p->terms, w_amax);
        rc = sgv_check_launch("conv3x3_prep_weights_pairs");
    } else {
        const int words = tiles_m(p->c_out) * (p->c_in / KC) * 9 * 2 * TM;
        hipLaunchKernelGGL(conv3x3_prep_weights, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, p->weight, (u32x4*)p->workspace, p->c_out, p->c_in, p->mode,
                           dtype == SGV_F16 ? 2 : p->terms, w_amax);
        rc = sgv_check_launch("conv3x3_prep_weights");
    }
    if (rc != SGV_OK) return rc;

    s2_params kp{};
    kp.x = (const float*)p->x; kp.wprep = (const u32x4*)p->workspace; kp.y = (float*)p->y;
    kp.n = p->n; kp.k = p->c_in; kp.m = p->c_out; kp.h = p->h; kp.w = p->w;
    if (p->terms == 4) { kp.x_amax = p->x_amax; kp.w_amax = w_amax; }
    kp.tiles = p->n * (p->h / S_ROWS) * (p->w / SEG) * (p->c_out / TM);
    kp.grid = std::min(kp.tiles, g_cus);
    const double small_px = (double)p->n * p->h * p->w, big_px = (double)p->n * (2 * p->h + 1) * (2 * p->w + 1);
    const double bytes = (io ? 2.0 : 4.0) * (p->mode == 0 ? big_px * p->c_in + small_px * p->c_out : small_px * p->c_in + big_px * p->c_out) + 4.0 * p->c_in * p->c_out * 9;
    sgv_launch_scope scope(SGV_K_CONV3X3, stream, bytes, 2.0 * small_px * p->c_in * (double)p->c_out * 9);
    if (pairs) {
        const int ss = p->w >= SEG ? 1 : 32 / p->w;
        kp.tiles = ss == 1 ? p->n * (p->h / P2_ROWS) * (p->w / SEG) * (p->c_out / P2_TM) : (p->n / ss) * (p->h / P2_ROWS) * (p->c_out / P2_TM);
        kp.grid = std::min(kp.tiles, g_cus);
        s2_epilogue ke{};
        if (ep) { ke.bias = ep->bias; ke.accumulate = ep->accumulate; ke.act_out = ep->act_out; ke.act = ep->act; ke.alpha = ep->alpha; ke.gain = ep->gain; ke.clamp = ep->clamp; }
#define SGV_P2_GO(T, E, SS, IO) hipLaunchKernelGGL((conv3x3_s2_pairs_kernel<T, 0, E, SS, IO>), dim3((unsigned)kp.grid), dim3(512), p2_lds_bytes(SS), stream, kp, ke)
#define SGV_P2_S(T, E, IO) do { if (ss == 1) SGV_P2_GO(T, E, 1, IO); else if (ss == 2) SGV_P2_GO(T, E, 2, IO); else SGV_P2_GO(T, E, 4, IO); } while (0)
        if (io == 1) { if (ep) SGV_P2_S(1, 1, 1); else SGV_P2_S(1, 0, 1); }
        else if (io == 2) { if (ep) SGV_P2_S(1, 1, 2); else SGV_P2_S(1, 0, 2); }
        else if (ep) { if (p->terms == 1) SGV_P2_S(1, 1, 0); else if (p->terms == 3) SGV_P2_S(3, 1, 0); else SGV_P2_S(4, 1, 0); }
        else { if (p->terms == 1) SGV_P2_S(1, 0, 0); else if (p->terms == 3) SGV_P2_S(3, 0, 0); else SGV_P2_S(4, 0, 0); }
#undef SGV_P2_S
#undef SGV_P2_GO
        if (io) sgv_note_variant(SGV_V_conv_s2_lowp);
        sgv_note_variant(ss == 1 ? (ep ? SGV_V_conv_s2_pairs_fused : SGV_V_conv_s2_pairs) : (ep ? SGV_V_conv_s2_pairs_packed_fused : SGV_V_conv_s2_pairs_packed));
        return sgv_check_launch("conv3x3_s2_pairs_kernel");
    }
    if (p->mode == 0 && g_s2_ws) {
        kp.tiles = p->n * (p->h / S2W_ROWS) * (p->w / SEG) * (p->c_out / TM);
        kp.grid = std::min(kp.tiles, g_cus);
        if (p->terms == 1) hipLaunchKernelGGL(conv3x3_s2_ws_kernel<1>, dim3((unsigned)kp.grid), dim3(512), S2W_LDS_BYTES, stream, kp);
        else if (p->terms == 3) hipLaunchKernelGGL(conv3x3_s2_ws_kernel<3>, dim3((unsigned)kp.grid), dim3(512), S2W_LDS_BYTES, stream, kp);
        else hipLaunchKernelGGL(conv3x3_s2_ws_kernel<4>, dim3((unsigned)kp.grid), dim3(512), S2W_LDS_BYTES, stream, kp);
        sgv_note_variant(SGV_V_conv_s2_ws);
        return sgv_check_launch("conv3x3_s2_ws_kernel");
    }
    if (p->mode == 0) {
        if (p->terms == 1) hipLaunchKernelGGL(conv3x3_s2_kernel<1>, dim3((unsigned)kp.grid), dim3(256), S_LDS_BYTES, stream, kp);
        else hipLaunchKernelGGL(conv3x3_s2_kernel<3>, dim3((unsigned)kp.grid), dim3(256), S_LDS_BYTES, stream, kp);
        sgv_note_variant(SGV_V_conv_s2_1role);
        return sgv_check_launch("conv3x3_s2_kernel");
    }
    if (io) {
        const int ss = packed ? 32 / p->w : 1;
        kp.tiles = ss == 1 ? p->n * (p->h / TW_ROWS) * (p->w / SEG) * tiles_m(p->c_out) : (p->n / ss) * (p->h / TW_ROWS) * (p->c_out / TM);
        kp.grid = std::min(kp.tiles, g_cus);
#define SGV_TW_GO(SS, IO) hipLaunchKernelGGL((convT3x3_s2_ws_kernel<1, 0, SS, IO>), dim3((unsigned)kp.grid), dim3(448), tw_lds_bytes(SS), stream, kp)
#define SGV_TW_S(IO) do { if (ss == 1) SGV_TW_GO(1, IO); else if (ss == 2) SGV_TW_GO(2, IO); else SGV_TW_GO(4, IO); } while (0)
        if (io == 1) SGV_TW_S(1); else SGV_TW_S(2);
#undef SGV_TW_S
#undef SGV_TW_GO
        sgv_note_variant(SGV_V_convT_lowp);
    } else if (packed) {
        const int ss = 32 / p->w;
        kp.tiles = (p->n / ss) * (p->h / TW_ROWS) * (p->c_out / TM);
        kp.grid = std::min(kp.tiles, g_cus);
#define SGV_TW_T(T, SS) hipLaunchKernelGGL((convT3x3_s2_ws_kernel<T, 0, SS>), dim3((unsigned)kp.grid), dim3(448), tw_lds_bytes(SS), stream, kp)
        if (ss == 2) { if (p->terms == 1) SGV_TW_T(1, 2); else if (p->terms == 3) SGV_TW_T(3, 2); else SGV_TW_T(4, 2); }
        else { if (p->terms == 1) SGV_TW_T(1, 4); else if (p->terms == 3) SGV_TW_T(3, 4); else SGV_TW_T(4, 4); }
    } else if (g_s2_ws) {
        kp.tiles = p->n * (p->h / TW_ROWS) * (p->w / SEG) * tiles_m(p->c_out);
        kp.grid = std::min(kp.tiles, g_cus);
        if (p->terms == 1) SGV_TW_T(1, 1); else if (p->terms == 3) SGV_TW_T(3, 1); else SGV_TW_T(4, 1);
#undef SGV_TW_T
    } else {
        if (p->terms == 1) hipLaunchKernelGGL(convT3x3_s2_kernel<1>, dim3((unsigned)kp.grid), dim3(512), T_LDS_BYTES, stream, kp);
        else hipLaunchKernelGGL(convT3x3_s2_kernel<3>, dim3((unsigned)kp.grid), dim3(512), T_LDS_BYTES, stream, kp);
    }
    sgv_note_variant(packed ? SGV_V_convT_ws_packed : g_s2_ws ? SGV_V_convT_ws : SGV_V_convT_1role);
    if (p->c_out % TM != 0) sgv_note_variant(SGV_V_convT_half_tile);
    rc = sgv_check_launch("convT3x3_s2_kernel");
    if (rc != SGV_OK) return rc;
    if (!g_edge_mfma) {
        float* edge = (float*)((char*)p->workspace + (size_t)p->c_in * p->c_out * 9 * 4);
        const size_t edge_floats = convT3x3_s2_edge_floats(p->n, p->c_in, p->c_out, p->h, p->w);
        hipLaunchKernelGGL(convT3x3_s2_edge_gather, dim3((unsigned)((edge_floats + 255) / 256)), dim3(256), 0, stream, (const float*)p->x, p->weight, edge, p->n, p->c_in, p->c_out,
                           p->h, p->w);
        const int lmax = std::max(2 * p->w + 1, 2 * p->h);
        hipLaunchKernelGGL(convT3x3_s2_edge_kernel, dim3((unsigned)((lmax + 127) / 128), (unsigned)(p->n * (p->c_out / EDGE_MC)), 2), dim3(128), 0, stream, edge, (float*)p->y, p->n,
                           p->c_in, p->c_out, p->h, p->w);
        sgv_note_variant(SGV_V_convT_edge_gather);
        return sgv_check_launch("convT3x3_s2_edge_kernel");
    }
    // last output row (oy = 2H) and column (ox = 2W): 0.4 % of the flops on the fp32 matrix pipe, after one pass that lays the six weight taps
    // and the last input column out contiguously
    {
        float* edge = (float*)((char*)p->workspace + (size_t)p->c_in * (tiles_m(p->c_out) * TM) * 10 * 4);
        const size_t prep = convT3x3_s2_edge_we_floats(p->c_in, p->c_out) + (size_t)p->n * p->c_in * p->h;
        const size_t prep_io = prep + (size_t)p->n * p->c_in * p->w;   // 16-bit tensors: the last input row is gathered as fp32 as well
        const dim3 eg((unsigned)((std::max(p->h, p->w) + 1 + 31) / 32), (unsigned)(p->n * (p->c_out / 32)), 2);
        // input channels split over the 4 (c_in % 32 == 0) or 2 waves of a workgroup; SGV_CONVT_EDGE_KSPLIT=0: one wave per block
        static const bool ksplit = !(getenv("SGV_CONVT_EDGE_KSPLIT") && getenv("SGV_CONVT_EDGE_KSPLIT")[0] == '0');
        const int nw = !ksplit ? 1 : (p->c_in % 32 == 0 ? 4 : 2);
#define SGV_EDGE_MFMA(IOV, XPTR) \
        if (nw == 4) hipLaunchKernelGGL((convT3x3_s2_edge_mfma<IOV, 4>), eg, dim3(256), 0, stream, XPTR, edge, p->y, p->n, p->c_in, p->c_out, p->h, p->w); \
        else if (nw == 2) hipLaunchKernelGGL((convT3x3_s2_edge_mfma<IOV, 2>), eg, dim3(128), 0, stream, XPTR, edge, p->y, p->n, p->c_in, p->c_out, p->h, p->w); \
        else hipLaunchKernelGGL((convT3x3_s2_edge_mfma<IOV, 1>), eg, dim3(64), 0, stream, XPTR, edge, p->y, p->n, p->c_in, p->c_out, p->h, p->w);
        if (io == 0) {
            hipLaunchKernelGGL(convT3x3_s2_edge_prep<0>, dim3((unsigned)((prep + 255) / 256)), dim3(256), 0, stream, p->x, p->weight, edge, p->n, p->c_in, p->c_out, p->h, p->w);
            SGV_EDGE_MFMA(0, (const float*)p->x)
        } else if (io == 1) {
            hipLaunchKernelGGL(convT3x3_s2_edge_prep<1>, dim3((unsigned)((prep_io + 255) / 256)), dim3(256), 0, stream, p->x, p->weight, edge, p->n, p->c_in, p->c_out, p->h, p->w);
            SGV_EDGE_MFMA(1, (const float*)nullptr)
        } else {
            hipLaunchKernelGGL(convT3x3_s2_edge_prep<2>, dim3((unsigned)((prep_io + 255) / 256)), dim3(256), 0, stream, p->x, p->weight, edge, p->n, p->c_in, p->c_out, p->h, p->w);
            SGV_EDGE_MFMA(2, (const float*)nullptr)
        }
#undef SGV_EDGE_MFMA
    }
    sgv_note_variant(SGV_V_convT_edge_mfma);
    return sgv_check_launch("convT3x3_s2_edge_mfma");
}

}  // namespace

extern "C" int sgv_conv3x3_s2(const sgv_conv3x3_params* p, int dtype, void* stream_) { return conv3x3_s2_impl(p, nullptr, dtype, stream_); }

extern "C" int sgv_conv3x3_s2_fused(const sgv_conv3x3_params* p, const sgv_conv3x3_s2_epilogue* e, int dtype, void* stream_) {
    if (!e) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_s2_fused: epilogue is NULL");
    return conv3x3_s2_impl(p, e, dtype, stream_);
}

extern "C" int sgv_conv3x3_s2_supported_mode(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int32_t mode, int dtype) {
    std::call_once(g_attr_once, init_once);
    return (supported_s2(n, c_in, c_out, h, w, dtype, mode) || (g_s2_ws && supported_s2_packed(n, c_in, c_out, h, w, mode, dtype))) && s2_mode_ok(c_out, h, mode, dtype) ? 1 : 0;
}

extern "C" int sgv_conv3x3_s2_fused_supported(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int dtype) {
    std::call_once(g_attr_once, init_once);
    return (supported_s2(n, c_in, c_out, h, w, dtype) || supported_s2_packed(n, c_in, c_out, h, w, 0, dtype)) && pairs_shape(c_out, h) && s2_mode_ok(c_out, h, 0, dtype) ? 1 : 0;
}
