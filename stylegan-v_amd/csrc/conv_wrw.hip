// C ABI entry of the 3x3 / stride-1 / pad-1 convolution weight gradient (kernel: wrw_kernel.h).
//
// Replaces, for the shapes it supports, the `Conv2dGradWeight` node of the reference's conv2d_gradfix
// (src/torch_utils/ops/conv2d_gradfix.py:140-170: `torch._C._jit_get_operation('aten::cudnn_convolution_backward_weight')`).

#include "sgv_common.h"
#include "wrw_kernel.h"
#include "wrw_ws_kernel.h"
#include "wrw_s2_ws_kernel.h"

#include <algorithm>
#include <mutex>
#include <stdlib.h>

using namespace sgv_wrw;

namespace {

// images 16 / 8 pixels wide: 2 / 4 samples per 32-pixel row step (producer / consumer kernel, PACK form)
bool packed_width(int h, int w) { return (w == 16 || w == 8) && h >= 1 && h <= 32; }

bool io16(int dtype) { return dtype == SGV_BF16 || dtype == SGV_F16; }

// 16-bit tensors (bf16 / fp16 dy and x, fp32 dw): the producer / consumer kernel on images >= 32 pixels wide, single bf16 operands (terms = 1)
bool supported(int n, int o, int i, int h, int w, int dtype) {
    if (!((dtype == SGV_F32 || io16(dtype)) && n >= 1 && o >= TO && i >= TI && o % TO == 0 && i % TI == 0 && h >= 1 && (int64_t)n * std::max(o, i) * h * w <= INT32_MAX)) return false;
    if (w >= SEG && w % SEG == 0 && (h <= 32 || h % 32 == 0)) return true;
    return dtype == SGV_F32 && packed_width(h, w);
}

bool supported_s2(int n, int cs, int cb, int h, int w, int dtype) {   // 16-bit tensors: the producer / consumer kernel (checked at launch: SGV_WRW_S2_WS)
    return (dtype == SGV_F32 || io16(dtype)) && n >= 1 && cs >= TO && cb >= TI && cs % TO == 0 && cb % TI == 0 && h >= 1 &&
           ((w >= SEG && w % SEG == 0 && (h <= 32 || h % 32 == 0)) || packed_width(h, w)) && (int64_t)n * std::max(cs, cb) * (2 * h + 1) * (2 * w + 1) <= INT32_MAX;
}

std::once_flag g_once, g_ws_once;
hipError_t g_attr_err = hipSuccess, g_ws_attr_err = hipSuccess;
bool g_use_s2_ws = true;   // SGV_WRW_S2_WS=0: the 4-wave stride-2 kernel of wrw_kernel.h instead of the producer / consumer form (wrw_s2_ws_kernel.h)
bool g_use_ws = true;   // SGV_WRW_WS=0: the 4-wave kernel of wrw_kernel.h instead of the producer / consumer form (wrw_ws_kernel.h)
int scatter_flush() {   // SGV_WRW_FLUSH=scatter: the element-per-lane atomics instead of the LDS-staged contiguous ones (wrw_kernel.h flush_tile)
    static const int v = [] { const char* e = getenv("SGV_WRW_FLUSH"); return (e && e[0] == 's') ? 1 : 0; }();
    return v;
}

void init_s2_once() {
    std::call_once(g_once, [] {
        hipError_t e = hipFuncSetAttribute((const void*)wrw3x3_s2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<1, false, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<1, false, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<1, true, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wrw3x3_s2_ws_kernel<1, true, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, WRW_S2_WS_LDS_BYTES);
        const char* env = getenv("SGV_WRW_S2_WS");
        g_use_s2_ws = !(env && env[0] == '0');
        g_attr_err = e;
    });
}

}  // namespace

extern "C" int sgv_conv3x3_wrw_supported(int32_t n, int32_t c_out, int32_t c_in, int32_t h, int32_t w, int dtype) {
    return supported(n, c_out, c_in, h, w, dtype) ? 1 : 0;
}

static int conv3x3_wrw_impl(const sgv_conv_wrw_params* p, const float* x_scale, int dtype, void* stream_) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw: params is NULL");
    if (!p->dy || !p->x || !p->dw) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw: NULL pointer");
    if (!supported(p->n, p->c_out, p->c_in, p->h, p->w, dtype))
        return sgv_fail(SGV_ERR_UNSUPPORTED, "conv3x3_wrw: needs fp32, channels %% 64 == 0, W %% 32 == 0 with H <= 32 or H %% 32 == 0, or W in {16, 8} with H <= 32 (got n=%d o=%d i=%d h=%d w=%d dtype=%d)",
                        p->n, p->c_out, p->c_in, p->h, p->w, dtype);
    if (p->terms != 1 && p->terms != 3 && p->terms != 4) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw: terms must be 1 (bf16 products), 3 (bf16 split) or 4 (block-scaled fp16 split)");
    if (io16(dtype) && p->terms != 1) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw: 16-bit tensors need terms = 1 (one 16-bit operand per value: bf16, or fp16 for fp16 tensors)");
    if (p->terms == 4 && (!p->dy_amax || !p->x_amax)) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw: terms = 4 needs dy_amax and x_amax, device pointers to upper bounds of max |dy| / max |x| (sgv_absmax)");
    if (p->terms == 4 && x_scale && !p->x_amax2)
        return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw_scaled: terms = 4 with x_scale needs x_amax2, a device pointer to an upper bound of max |x_scale| (the operand is x * x_scale: "
                                             "scaled by the bound of x alone, any |x_scale| > 2 overflows the fp16 split)");
    if ((((uintptr_t)p->dy) | ((uintptr_t)p->x)) & 15) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw: dy and x must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    wrw_params kp{};
    kp.dy = (const float*)p->dy; kp.x = (const float*)p->x; kp.dw = p->dw;
    kp.n = p->n; kp.o = p->c_out; kp.i = p->c_in; kp.h = p->h; kp.w = p->w;
    kp.xscale = x_scale;
    if (p->terms == 4) { kp.dy_amax = p->dy_amax; kp.x_amax = p->x_amax; kp.x_amax2 = x_scale ? p->x_amax2 : nullptr; }
    kp.scatter_flush = scatter_flush();
    kp.tiles_i = p->c_in / TI;
    const int tiles = (p->c_out / TO) * kp.tiles_i;
    // One workgroup per CU (profiles/r01_wrw_lab_v1.log: 256 persistent workgroups beat 512), spread over the output tiles.
    const int max_splits = std::max(1, 256 / std::max(1, std::min(tiles, 256)));
    // Rows per unit: every unit pays a two-barrier prologue with exposed load latency, so take the tallest row block that still spreads evenly
    // over the workgroups (whole column segments where the batch allows it).
    const bool pack = p->w < SEG;
    const int columns = pack ? (p->n + SEG / p->w - 1) / (SEG / p->w) : p->n * (p->w / SEG);   // 32-pixel column segments of the batch (packed: sample groups)
    kp.rows = std::min(p->h, 32);
    for (int rows : {p->h, 64}) {
        if (rows > p->h || p->h % rows) continue;
        const int units = columns * (p->h / rows), splits = std::min(units, max_splits);
        if (units % splits == 0 || units >= 8 * splits) { kp.rows = rows; break; }
    }
    kp.units = columns * (p->h / kp.rows);
    kp.splits = std::max(1, std::min(kp.units, max_splits));
    const size_t dw_bytes = (size_t)p->c_out * p->c_in * 9 * sizeof(float);
    hipError_t e = hipMemsetAsync(p->dw, 0, dw_bytes, stream);
    if (e != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "conv3x3_wrw: hipMemsetAsync failed: %s", hipGetErrorString(e));
    const double elems = (double)p->n * p->h * p->w;
    sgv_launch_scope scope(SGV_K_CONV_WRW, stream, (io16(dtype) ? 2.0 : 4.0) * elems * (p->c_out + p->c_in) + dw_bytes, 2.0 * elems * p->c_out * (double)p->c_in * 9);
    dim3 grid((unsigned)tiles, (unsigned)kp.splits);
    std::call_once(g_ws_once, [] {
        hipError_t e2 = hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(1));
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(1));
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<1, 1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(1));
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<3, 1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(1));
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(1));
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<4, 1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(1));
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<1, 1, 0, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(1));
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)wrw3x3_ws_kernel<1, 1, 0, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, wrw_ws_lds_bytes(1));
        g_ws_attr_err = e2;
        const char* env = getenv("SGV_WRW_WS");
        g_use_ws = !(env && env[0] == '0');
    });
    if (g_ws_attr_err != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "conv3x3_wrw: hipFuncSetAttribute failed: %s", hipGetErrorString(g_ws_attr_err));
    if (io16(dtype)) {
        if (dtype == SGV_BF16) hipLaunchKernelGGL((wrw3x3_ws_kernel<1, 1, 0, false, 1>), grid, dim3(512), wrw_ws_lds_bytes(1), stream, kp);
        else hipLaunchKernelGGL((wrw3x3_ws_kernel<1, 1, 0, false, 2>), grid, dim3(512), wrw_ws_lds_bytes(1), stream, kp);
        sgv_note_variant(SGV_V_wrw_lowp);
        return sgv_check_launch("wrw3x3_ws_kernel (16-bit tensors)");
    }
    if (pack) {   // the packed form exists in the producer / consumer kernel only
        if (p->terms == 1) hipLaunchKernelGGL((wrw3x3_ws_kernel<1, 1, 0, true>), grid, dim3(512), wrw_ws_lds_bytes(1), stream, kp);
        else if (p->terms == 3) hipLaunchKernelGGL((wrw3x3_ws_kernel<3, 1, 0, true>), grid, dim3(512), wrw_ws_lds_bytes(1), stream, kp);
        else hipLaunchKernelGGL((wrw3x3_ws_kernel<4, 1, 0, true>), grid, dim3(512), wrw_ws_lds_bytes(1), stream, kp);
        sgv_note_variant(SGV_V_wrw_s1_ws_packed);
        return sgv_check_launch("wrw3x3_ws_kernel (packed)");
    }
    if (g_use_ws || p->terms == 4) {      // (the block-scaled split: the producer / consumer kernel only)
        if (p->terms == 1) hipLaunchKernelGGL((wrw3x3_ws_kernel<1, 1>), grid, dim3(512), wrw_ws_lds_bytes(1), stream, kp);
        else if (p->terms == 3) hipLaunchKernelGGL((wrw3x3_ws_kernel<3, 1>), grid, dim3(512), wrw_ws_lds_bytes(1), stream, kp);
        else hipLaunchKernelGGL((wrw3x3_ws_kernel<4, 1>), grid, dim3(512), wrw_ws_lds_bytes(1), stream, kp);
        sgv_note_variant(x_scale ? SGV_V_wrw_s1_ws_scaled : SGV_V_wrw_s1_ws);
        return sgv_check_launch("wrw3x3_ws_kernel");
    }
    if (x_scale) return sgv_fail(SGV_ERR_UNSUPPORTED, "conv3x3_wrw_scaled: the 4-wave kernel (SGV_WRW_WS=0) has no input scale");
    if (p->terms == 1) hipLaunchKernelGGL(wrw3x3_kernel<1>, grid, dim3(256), 0, stream, kp);
    else hipLaunchKernelGGL(wrw3x3_kernel<3>, grid, dim3(256), 0, stream, kp);
    sgv_note_variant(SGV_V_wrw_s1_4wave);
    return sgv_check_launch("wrw3x3_kernel");
}

extern "C" int sgv_conv3x3_wrw(const sgv_conv_wrw_params* p, int dtype, void* stream_) { return conv3x3_wrw_impl(p, nullptr, dtype, stream_); }

extern "C" int sgv_conv3x3_wrw_scaled(const sgv_conv_wrw_params* p, const float* x_scale, int dtype, void* stream_) {
    if (!x_scale) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw_scaled: x_scale is NULL");
    return conv3x3_wrw_impl(p, x_scale, dtype, stream_);
}

extern "C" int sgv_conv3x3_wrw_s2_supported(int32_t n, int32_t c_small, int32_t c_big, int32_t h, int32_t w, int dtype) {
    if (io16(dtype)) { init_s2_once(); if (!g_use_s2_ws) return 0; }   // 16-bit tensors: the producer / consumer kernel only
    return supported_s2(n, c_small, c_big, h, w, dtype) ? 1 : 0;
}

// Stride-2 member: p->dy is the SMALL tensor [n, c_out, h, w], p->x the BIG one [n, c_in, 2h+1, 2w+1]; dw is [c_out, c_in, 3, 3].
extern "C" int sgv_conv3x3_wrw_s2(const sgv_conv_wrw_params* p, int dtype, void* stream_) {
    if (!p) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw_s2: params is NULL");
    if (!p->dy || !p->x || !p->dw) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw_s2: NULL pointer");
    if (!supported_s2(p->n, p->c_out, p->c_in, p->h, p->w, dtype))
        return sgv_fail(SGV_ERR_UNSUPPORTED, "conv3x3_wrw_s2: needs fp32, channels %% 64 == 0, and on the small HxW grid W %% 32 == 0 with H <= 32 or H %% 32 == 0, or W in {16, 8} with H <= 32 (got n=%d cs=%d cb=%d h=%d w=%d dtype=%d)",
                        p->n, p->c_out, p->c_in, p->h, p->w, dtype);
    if (p->terms != 1 && p->terms != 3 && p->terms != 4) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw_s2: terms must be 1, 3 or 4");
    if (p->terms == 4 && (!p->dy_amax || !p->x_amax)) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw_s2: terms = 4 needs dy_amax and x_amax, device pointers to upper bounds of max |dy| / max |x| (sgv_absmax)");
    if (io16(dtype) && p->terms != 1) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw_s2: 16-bit tensors need terms = 1 (one 16-bit operand per value: bf16, or fp16 for fp16 tensors)");
    if (((uintptr_t)p->dy) & (io16(dtype) ? 7 : 15)) return sgv_fail(SGV_ERR_INVALID_ARG, "conv3x3_wrw_s2: the small tensor must be aligned to four elements");
    init_s2_once();
    if (g_attr_err != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "conv3x3_wrw_s2: hipFuncSetAttribute failed: %s", hipGetErrorString(g_attr_err));
    if (io16(dtype) && !g_use_s2_ws) return sgv_fail(SGV_ERR_UNSUPPORTED, "conv3x3_wrw_s2: 16-bit tensors need the producer / consumer kernel (SGV_WRW_S2_WS != 0)");
    hipStream_t stream = (hipStream_t)stream_;
    wrw_s2_params kp{};
    kp.small = (const float*)p->dy; kp.big = (const float*)p->x; kp.dw = p->dw;
    kp.n = p->n; kp.cs = p->c_out; kp.cb = p->c_in; kp.h = p->h; kp.w = p->w;
    kp.rows = std::min(p->h, 32);
    if (p->terms == 4) { kp.small_amax = p->dy_amax; kp.big_amax = p->x_amax; }
    kp.scatter_flush = scatter_flush();
    kp.tiles_b = p->c_in / TI;
    const bool pack = p->w < SEG;
    kp.units = (pack ? (p->n + SEG / p->w - 1) / (SEG / p->w) : p->n * (p->w / SEG)) * (p->h / kp.rows);
    const int tiles = (p->c_out / TO) * kp.tiles_b;
    kp.splits = std::max(1, std::min(kp.units, 256 / std::max(1, std::min(tiles, 256))));
    const size_t dw_bytes = (size_t)p->c_out * p->c_in * 9 * sizeof(float);
    hipError_t e = hipMemsetAsync(p->dw, 0, dw_bytes, stream);
    if (e != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "conv3x3_wrw_s2: hipMemsetAsync failed: %s", hipGetErrorString(e));
    const double small_px = (double)p->n * p->h * p->w, big_px = (double)p->n * (2 * p->h + 1) * (2 * p->w + 1);
    sgv_launch_scope scope(SGV_K_CONV_WRW, stream, (io16(dtype) ? 2.0 : 4.0) * (small_px * p->c_out + big_px * p->c_in) + dw_bytes, 2.0 * small_px * p->c_out * (double)p->c_in * 9);
    dim3 grid((unsigned)tiles, (unsigned)kp.splits);
    if (io16(dtype)) {
        if (pack) {
            if (dtype == SGV_BF16) hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<1, true, 0, 1>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
            else hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<1, true, 0, 2>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
        } else {
            if (dtype == SGV_BF16) hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<1, false, 0, 1>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
            else hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<1, false, 0, 2>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
        }
        sgv_note_variant(SGV_V_wrw_s2_lowp);
        return sgv_check_launch("wrw3x3_s2_ws_kernel (16-bit tensors)");
    }
    if (g_use_s2_ws || p->terms == 4) {
        if (pack) {
            if (p->terms == 1) hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<1, true>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
            else if (p->terms == 3) hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<3, true>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
            else hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<4, true>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
        } else {
            if (p->terms == 1) hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<1, false>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
            else if (p->terms == 3) hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<3, false>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
            else hipLaunchKernelGGL((wrw3x3_s2_ws_kernel<4, false>), grid, dim3(512), WRW_S2_WS_LDS_BYTES, stream, kp);
        }
        sgv_note_variant(pack ? SGV_V_wrw_s2_ws_packed : SGV_V_wrw_s2_ws);
        return sgv_check_launch("wrw3x3_s2_ws_kernel");
    }
    if (pack) {
        if (p->terms == 1) hipLaunchKernelGGL((wrw3x3_s2_kernel<1, true>), grid, dim3(256), WRW_S2_LDS_BYTES, stream, kp);
        else hipLaunchKernelGGL((wrw3x3_s2_kernel<3, true>), grid, dim3(256), WRW_S2_LDS_BYTES, stream, kp);
        sgv_note_variant(SGV_V_wrw_s2_4wave_packed);
        return sgv_check_launch("wrw3x3_s2_kernel (packed)");
    }
    if (p->terms == 1) hipLaunchKernelGGL(wrw3x3_s2_kernel<1>, grid, dim3(256), WRW_S2_LDS_BYTES, stream, kp);
    else hipLaunchKernelGGL(wrw3x3_s2_kernel<3>, grid, dim3(256), WRW_S2_LDS_BYTES, stream, kp);
    sgv_note_variant(SGV_V_wrw_s2_4wave);
    return sgv_check_launch("wrw3x3_s2_kernel");
}
