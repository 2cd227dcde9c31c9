/*
 * sgv_oracle.c -- TEST INFRASTRUCTURE ONLY.  Scalar CPU restatement of the two native kernels
 * of the universome/stylegan-v hot path.  Nothing in the product (stylegan-v_amd/) may import,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Each function follows the reference's own scalar algorithm, line by line:
 *   oracle_upfirdn2d  <- upfirdn2d_kernel_large, src/torch_utils/ops/upfirdn2d.cu:29-92
 *                        (floor_div :20-24, receptive field :43-48,60-65, tap walk :68-89),
 *                        output size src/torch_utils/ops/upfirdn2d.cpp:32-33
 *   oracle_bias_act   <- bias_act_kernel, src/torch_utils/ops/bias_act.cu:23-147
 *                        (bias index (xi / stepB) % sizeB :44, formulas :51-142)
 * The reference's native code is CUDA and cannot be built here (no nvcc), so the restatement is
 * pinned instead against the reference's own Python fallbacks (`_upfirdn2d_ref`, `_bias_act_ref`,
 * and autograd through them) via the golden vectors in tests/golden/ -- see tests/test_oracle.py.
 *
 * Accumulation: v = fma(x, f, v) in the reference's loop order (rows, then columns), fp32 for
 * fp32/fp16/bf16 data and fp64 for fp64 -- the internal type of upfirdn2d.cu:15-18.  The HIP kernels
 * use the same order and explicit fma, so HIP == oracle bit for bit (not merely within tolerance).
 * 16-bit types are carried as uint16_t and converted with round-to-nearest-even.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2, DT_F64 = 3 };

/* ---- 16-bit float conversions (IEEE binary16 and bfloat16), round-to-nearest-even ---- */
static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { /* subnormal */
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
static uint16_t float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t abs = x & 0x7fffffffu;
    if (abs > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);       /* NaN */
    if (abs >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);      /* overflow -> inf (>= 65520) */
    if (abs < 0x33000001u) return (uint16_t)sign;                   /* underflow to zero (<= 2^-25) */
    int32_t exp = (int32_t)(abs >> 23) - 127;
    uint32_t man = (abs & 0x7fffffu) | 0x800000u;
    if (exp < -14) { /* subnormal half */
        int shift = -14 - exp + 13; /* bits to drop */
        uint32_t half_man = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man++;
        return (uint16_t)(sign | half_man);
    }
    uint32_t half = ((uint32_t)(exp + 15) << 10) | ((man >> 13) & 0x3ffu);
    uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;   /* may carry into exponent: correct */
    return (uint16_t)(sign | half);
}
static float bf16_to_float(uint16_t b) { uint32_t x = (uint32_t)b << 16; float f; memcpy(&f, &x, 4); return f; }
static uint16_t float_to_bf16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

static double load_elem(const void* p, int64_t i, int dt) {
    switch (dt) {
        case DT_F32: return ((const float*)p)[i];
        case DT_F16: return half_to_float(((const uint16_t*)p)[i]);
        case DT_BF16: return bf16_to_float(((const uint16_t*)p)[i]);
        default: return ((const double*)p)[i];
    }
}
static void store_elem(void* p, int64_t i, int dt, double v) {
    switch (dt) {
        case DT_F32: ((float*)p)[i] = (float)v; break;
        case DT_F16: ((uint16_t*)p)[i] = float_to_half((float)v); break;
        case DT_BF16: ((uint16_t*)p)[i] = float_to_bf16((float)v); break;
        default: ((double*)p)[i] = v; break;
    }
}

static int floor_div(int a, int b) { /* upfirdn2d.cu:20-24 */
    int t = 1 - a / b;
    return (a + t * b) / b - t;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* Output extent, upfirdn2d.cpp:32-33 (C integer division). */
int oracle_upfirdn2d_out_size(int in_size, int up, int down, int pad0, int pad1, int taps) {
    return (in_size * up + pad0 + pad1 - taps + down) / down;
}

/*
 * x: [n, c, in_h, in_w] with element strides (in_sn, in_sc, in_sh, in_sw); f: fp32 [f_h, f_w] with
 * strides (f_sh, f_sw); y likewise.  Returns 0, or -1 if the derived output size is < 1.
 */
int oracle_upfirdn2d(const void* x, const float* f, void* y, int dt,
                     int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_x1, int pad_y0, int pad_y1, int flip, float gain,
                     int in_w, int in_h, int in_c, int in_n,
                     int64_t in_sw, int64_t in_sh, int64_t in_sc, int64_t in_sn,
                     int f_w, int f_h, int64_t f_sw, int64_t f_sh,
                     int64_t out_sw, int64_t out_sh, int64_t out_sc, int64_t out_sn) {
    const int out_w = oracle_upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, f_w);
    const int out_h = oracle_upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, f_h);
    if (out_w < 1 || out_h < 1) return -1;
    for (int n = 0; n < in_n; n++)
    for (int c = 0; c < in_c; c++)
    for (int out_y = 0; out_y < out_h; out_y++) {
        /* Y receptive field, upfirdn2d.cu:43-48 */
        int mid_y = out_y * down_y + up_y - 1 - pad_y0;
        int in_y = imin(imax(floor_div(mid_y, up_y), 0), in_h);
        int h = imin(imax(floor_div(mid_y + f_h, up_y), 0), in_h) - in_y;
        int filter_y = mid_y + f_h - (in_y + 1) * up_y;
        if (flip) filter_y = f_h - 1 - filter_y;
        for (int out_x = 0; out_x < out_w; out_x++) {
            /* X receptive field, upfirdn2d.cu:60-65 */
            int mid_x = out_x * down_x + up_x - 1 - pad_x0;
            int in_x = imin(imax(floor_div(mid_x, up_x), 0), in_w);
            int w = imin(imax(floor_div(mid_x + f_w, up_x), 0), in_w) - in_x;
            int filter_x = mid_x + f_w - (in_x + 1) * up_x;
            if (flip) filter_x = f_w - 1 - filter_x;
            int64_t xo = in_x * in_sw + in_y * in_sh + c * in_sc + n * in_sn;
            int64_t fo = filter_x * f_sw + filter_y * f_sh;
            int64_t step_x = (flip ? up_x : -up_x) * f_sw;
            int64_t step_y = (flip ? up_y : -up_y) * f_sh;
            /* Inner loop, upfirdn2d.cu:74-88: rows outer, columns inner, one fma per tap */
            if (dt == DT_F64) {
                double v = 0;
                for (int yy = 0; yy < h; yy++)
                    for (int xx = 0; xx < w; xx++)
                        v = fma(load_elem(x, xo + yy * in_sh + xx * in_sw, dt), (double)f[fo + yy * step_y + xx * step_x], v);
                v *= (double)gain;
                store_elem(y, out_x * out_sw + out_y * out_sh + c * out_sc + n * out_sn, dt, v);
            } else {
                float v = 0;
                for (int yy = 0; yy < h; yy++)
                    for (int xx = 0; xx < w; xx++)
                        v = fmaf((float)load_elem(x, xo + yy * in_sh + xx * in_sw, dt), f[fo + yy * step_y + xx * step_x], v);
                v *= gain;
                store_elem(y, out_x * out_sw + out_y * out_sh + c * out_sc + n * out_sn, dt, v);
            }
        }
    }
    return 0;
}

/* ---- bias_act, one element in scalar type S; macro-instantiated for float and double ---- */
#define DEFINE_BA_EVAL(NAME, S, EXP, LOG)                                                                     \
static S NAME(int A, int G, S x, S b, S xref, S yref, S dy, S alpha, S gain, S clamp) {                       \
    const S one = 1, two = 2, expRange = 80, halfExpRange = 40;                                               \
    const S seluScale = (S)1.0507009873554804934193349852946;                                                 \
    const S seluAlpha = (S)1.6732632423543772848170429916717;                                                 \
    S yy = (gain != 0) ? yref / gain : 0;                                                                     \
    S y = 0;                                                                                                  \
    if (G == 0) x += b; else xref += b;                       /* bias_act.cu:51 */                            \
    if (A == 1) { if (G == 0) y = x; if (G == 1) y = x; }                                                     \
    if (A == 2) { if (G == 0) y = (x > 0) ? x : 0; if (G == 1) y = (yy > 0) ? x : 0; }                        \
    if (A == 3) { if (G == 0) y = (x > 0) ? x : x * alpha; if (G == 1) y = (yy > 0) ? x : x * alpha; }        \
    if (A == 4) {                                                                                             \
        if (G == 0) { S c = EXP(x); S d = one / c; y = (x < -expRange) ? -one : (x > expRange) ? one : (c - d) / (c + d); } \
        if (G == 1) y = x * (one - yy * yy);                                                                  \
        if (G == 2) y = x * (one - yy * yy) * (-two * yy);                                                    \
    }                                                                                                         \
    if (A == 5) {                                                                                             \
        if (G == 0) y = (x < -expRange) ? 0 : one / (EXP(-x) + one);                                          \
        if (G == 1) y = x * yy * (one - yy);                                                                  \
        if (G == 2) y = x * yy * (one - yy) * (one - two * yy);                                               \
    }                                                                                                         \
    if (A == 6) {                                                                                             \
        if (G == 0) y = (x >= 0) ? x : EXP(x) - one;                                                          \
        if (G == 1) y = (yy >= 0) ? x : x * (yy + one);                                                       \
        if (G == 2) y = (yy >= 0) ? 0 : x * (yy + one);                                                       \
    }                                                                                                         \
    if (A == 7) {                                                                                             \
        if (G == 0) y = (x >= 0) ? seluScale * x : (seluScale * seluAlpha) * (EXP(x) - one);                  \
        if (G == 1) y = (yy >= 0) ? x * seluScale : x * (yy + seluScale * seluAlpha);                         \
        if (G == 2) y = (yy >= 0) ? 0 : x * (yy + seluScale * seluAlpha);                                     \
    }                                                                                                         \
    if (A == 8) {                                                                                             \
        if (G == 0) y = (x > expRange) ? x : LOG(EXP(x) + one);                                               \
        if (G == 1) y = x * (one - EXP(-yy));                                                                 \
        if (G == 2) { S c = EXP(-yy); y = x * c * (one - c); }                                                \
    }                                                                                                         \
    if (A == 9) {                                                                                             \
        if (G == 0) y = (x < -expRange) ? 0 : x / (EXP(-x) + one);                                            \
        else {                                                                                                \
            S c = EXP(xref); S d = c + one;                                                                   \
            if (G == 1) y = (xref > halfExpRange) ? x : x * c * (xref + d) / (d * d);                         \
            else y = (xref > halfExpRange) ? 0 : x * c * (xref * (two - d) + two * d) / (d * d * d);          \
            yref = (xref < -expRange) ? 0 : xref / (EXP(-xref) + one) * gain;                                 \
        }                                                                                                     \
    }                                                                                                         \
    y *= gain * dy;                                           /* bias_act.cu:133 */                           \
    if (clamp >= 0) {                                         /* bias_act.cu:136-142 */                       \
        if (G == 0) y = (y > -clamp && y < clamp) ? y : (y >= 0) ? clamp : -clamp;                            \
        else y = (yref > -clamp && yref < clamp) ? y : 0;                                                     \
    }                                                                                                         \
    return y;                                                                                                 \
}
DEFINE_BA_EVAL(ba_eval_f, float, expf, logf)
DEFINE_BA_EVAL(ba_eval_d, double, exp, log)

/* All tensors share one dense layout of size_x elements; b/xref/yref/dy may be NULL. */
int oracle_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                    int dt, int grad, int act, float alpha, float gain, float clamp,
                    int64_t size_x, int64_t size_b, int64_t step_b) {
    if (act < 1 || act > 9 || grad < 0 || grad > 2) return -1;
    for (int64_t xi = 0; xi < size_x; xi++) {
        double xv = load_elem(x, xi, dt);
        double bv = b ? load_elem(b, (xi / step_b) % size_b, dt) : 0;     /* bias_act.cu:44 */
        double xr = xref ? load_elem(xref, xi, dt) : 0;
        double yr = yref ? load_elem(yref, xi, dt) : 0;
        double dv = dy ? load_elem(dy, xi, dt) : 1;
        double out;
        if (dt == DT_F64) out = ba_eval_d(act, grad, xv, bv, xr, yr, dv, (double)alpha, (double)gain, (double)clamp);
        else out = ba_eval_f(act, grad, (float)xv, (float)bv, (float)xr, (float)yr, (float)dv, alpha, gain, clamp);
        store_elem(y, xi, dt, out);
    }
    return 0;
}
