/*
 * sgv_ops.h -- C ABI of libsgv_hip.so, the MI355X (gfx950) kernel library behind the
 * StyleGAN-V synthesis/discriminator op stack.
 *
 * Every entry point is plain C: raw device pointers, sizes, strides, a HIP stream passed as
 * void*.  No torch types cross this boundary.  The library owns nothing: callers allocate
 * inputs/outputs (the Python host layer uses torch's caching allocator) and pass the stream
 * the work must be ordered on.  No entry point allocates device memory or synchronises, so
 * every launch is hipGraph-capturable.  All functions return 0 on success or a negative
 * SGV_ERR_* code; sgv_last_error() returns a thread-local message for the last failure.
 * Nothing throws across the ABI.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * universome/stylegan-v checkout); the comment in front of each declaration below is the authority,
 * this map is the index:
 *
 *   sgv_upfirdn2d, sgv_upfirdn2d_kernel_kind
 *                      <- `_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)`
 *                          src/torch_utils/ops/upfirdn2d.cpp:16,98-101 (parameter block upfirdn2d.h:14-40, kernels upfirdn2d.cu:29-200)
 *   sgv_upfirdn2d_fused <- no single reference op: upfirdn2d -> x*dcoefs -> bias_act of an up-sampling SynthesisLayer
 *                          (src/training/networks.py:65-74,141-143; modes 1, 2), the gradient of "bias_act, then the FIR in front of a
 *                          strided convolution" (DiscriminatorBlock conv0 -> conv1, networks.py:343-344; mode 3), FIR gradient + add (mode 4)
 *   sgv_bias_act, sgv_bias_act_db
 *                      <- `_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)` src/torch_utils/ops/bias_act.cpp:32,94-97
 *                          (parameter block bias_act.h:12-31, kernel bias_act.cu:23-147); `_db`: with `dx.sum(...)` of bias_act.py:185
 *   sgv_weight_sqsum, sgv_demod_coefs, sgv_scale_channels, sgv_plane_dot
 *                      <- the weight (de)modulation arithmetic of `modulated_conv2d` networks.py:57-74 and its styles gradient
 *                          (no native counterpart in the reference: it materialises w[N,O,I,kh,kw] in PyTorch)
 *   sgv_act_grad_scale[_t], sgv_scale_dot[_t]
 *                      <- the element-wise backward of a fused stride-1 layer: bias_act grad-1 (bias_act.cu:60-61,133-142) x dcoefs with the
 *                          bias / dcoefs plane sums; `dxs * styles` with the styles gradient (networks.py:66)
 *   sgv_pointwise_small, sgv_pointwise_outer, sgv_pointwise_act, sgv_pointwise_small_gradin, sgv_pointwise_outer_act
 *                      <- the 1x1 `conv2d` of ToRGBLayer (networks.py:148-163, C_out = 3) and of the discriminator's `fromrgb`
 *                          layer (networks.py:447, C_in = 3), their gradients (conv2d_resample.py:40-54), and fromRGB with its
 *                          `bias_act` (layers.py Conv2dLayer.forward) as one pass in each direction
 *   sgv_conv3x3, sgv_conv3x3_supported, sgv_conv3x3_workspace_bytes
 *                      <- `conv2d` / `conv_transpose2d` of conv2d_gradfix.py:35-43,100-118 for 3x3 stride-1 layers (forward + data gradient)
 *   sgv_conv3x3_fused, sgv_conv3x3_fused_supported
 *                      <- a whole stride-1 layer: x*styles -> conv -> *dcoefs -> bias_act (networks.py:65-74,141-143; Conv2dLayer.forward)
 *   sgv_conv3x3_s2, sgv_conv3x3_s2_supported, sgv_conv3x3_s2_supported_mode, sgv_conv3x3_s2_workspace_bytes
 *                      <- the stride-2 `conv2d` / `conv_transpose2d` either side of the FIR, conv2d_resample.py:113-137
 *   sgv_conv3x3_s2_fused, sgv_conv3x3_s2_fused_supported
 *                      <- a down-sampling layer's tail: strided conv -> bias_act -> `y.add_(x)` of DiscriminatorBlock.forward networks.py:343-345
 *   sgv_conv3x3_wrw, sgv_conv3x3_wrw_scaled, sgv_conv3x3_wrw_s2 (+ _supported)
 *                      <- `Conv2dGradWeight` of conv2d_gradfix.py:140-170 (cudnn_convolution_backward_weight) for all of the above
 *   sgv_gemm_f32       <- the dense 1x1 skip `conv2d` of DiscriminatorBlock (networks.py:452 via conv2d_resample.py:40-54), its data and
 *                          weight gradients; the unfolded trajectory convolutions of motion.py:18-156 at thousands of rows
 *   sgv_fc             <- `FullyConnectedLayer.forward` src/training/layers.py:108-138 (weight gain, bias gain, addmm / matmul, bias_act),
 *                          `normalize_2nd_moment` layers.py:16-18, and through unfolded rows `EqLRConv1d` of motion.py
 *   sgv_multi_nan_to_num_f32
 *                      <- the `misc.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5, out=param.grad)` loop of
 *                          src/training/training_loop.py:384-386 as one launch over the whole gradient list
 *   sgv_multi_scale_f32 <- the equalised learning-rate products `self.weight * (self.weight_gain * self.lr_multiplier)` /
 *                          `self.bias * self.lr_multiplier` of every Conv2dLayer of a module (src/training/layers.py:184-185) and of their gradients
 *   sgv_time_encode    <- `AlignedTimeEncoder.forward` element-wise tail src/training/motion.py:201-212
 *   sgv_affine_resample <- `affine_grid` + `grid_sample` of the ADA geometric execution src/training/augment.py:297-300 and its backward
 *   sgv_ada_geometric  <- reflect pad + upsample2d + affine_grid / grid_sample + downsample2d of augment.py:270-300 in one pass (forward)
 *   sgv_ada_geometric_adjoint
 *                      <- the backward of that block as one pass: autograd through F.pad / upfirdn2d.py:249-260 / grid_sample_gradfix.py:45-83
 *                          (loss.py:91-110 differentiates through augmented fakes, :144-164 twice through augmented reals)
 *   sgv_prof_*, sgv_launch_count, sgv_variant_count, sgv_variant_name
 *                      <- no reference counterpart: per-launch HIP-event timing and launch / kernel-variant counters used by
 *                          bench.py (roofline numbers) and by the tests (proof of which kernel served a shape)
 *   sgv_version, sgv_last_error
 *                      <- the TORCH_CHECK messages of upfirdn2d.cpp:19-36 / bias_act.cpp:35-81 as a thread-local string
 */
#ifndef SGV_OPS_H
#define SGV_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGV_VERSION 105 /* major*100 + minor */

/* element types (the reference dispatches double/float/half: upfirdn2d.cpp:59, bias_act.cpp:76;
 * bf16 is this library's extension) */
enum sgv_dtype { SGV_F32 = 0, SGV_F16 = 1, SGV_BF16 = 2, SGV_F64 = 3 };

enum sgv_error {
    SGV_OK = 0,
    SGV_ERR_INVALID_ARG = -1, /* precondition of the reference's TORCH_CHECKs violated */
    SGV_ERR_TOO_LARGE = -2,   /* numel > INT_MAX (upfirdn2d.cpp:22-23,36; bias_act.cpp:40) */
    SGV_ERR_UNSUPPORTED = -3, /* dtype / activation index not implemented */
    SGV_ERR_LAUNCH = -4       /* hipLaunchKernel failed */
};

/* ---------------------------------------------------------------------------------------
 * upfirdn2d: pad -> zero-insert (up) -> FIR -> decimate (down).
 * Sizes/strides are in ELEMENTS, ordered [w, h, c, n] exactly like upfirdn2d.h:25-30.
 * out_w = (in_w*up_x + pad_x0 + pad_x1 - f_w + down_x) / down_x  (C division), same for h
 * (upfirdn2d.cpp:32-33).  The caller computes it and allocates y; the library re-derives it
 * and rejects a mismatch.
 */
typedef struct sgv_upfirdn2d_params {
    const void* x;  /* [n, c, in_h, in_w] with arbitrary dense strides */
    const float* f; /* [f_h, f_w] fp32, device memory */
    void* y;        /* [n, c, out_h, out_w] */
    int32_t up_x, up_y;
    int32_t down_x, down_y;
    int32_t pad_x0, pad_x1, pad_y0, pad_y1;
    int32_t flip; /* 0: true convolution, 1: correlation (upfirdn2d.py:153) */
    float gain;
    int32_t in_w, in_h, in_c, in_n;
    int64_t in_sw, in_sh, in_sc, in_sn; /* x strides */
    int32_t f_w, f_h;
    int64_t f_sw, f_sh; /* filter strides */
    int32_t out_w, out_h;
    int64_t out_sw, out_sh, out_sc, out_sn; /* y strides */
} sgv_upfirdn2d_params;

int sgv_upfirdn2d(const sgv_upfirdn2d_params* p, int dtype, void* stream);

/* Fused "FIR + scale + bias + activation + clamp" (the honest reading of north_star's `filtered_lrelu`: the
 * reference has no such op; it runs upfirdn2d -> x*dcoefs -> bias_act as three passes after every up-sampling
 * synthesis convolution, networks.py:65-74,141-143 + conv2d_resample.py:138-139).
 *   mode 1  y  = clamp(act(upfirdn2d(x) * scale[n,c] + bias[c]) * gain)                    (forward)
 *   mode 2  dx = upfirdn2d(g * scale[n,c]),  g = dy * act'(.) * gain masked by the clamp,   (backward of mode 1; the
 *           x = dy, yref = the forward output; additionally sum_g[n,c] += sum(g) and         upfirdn2d params describe the
 *           sum_gv[n,c] += sum(g * preactivation) over every plane, from which                transposed FIR, i.e. pad 2)
 *           dbias[c] = sum_n sum_g[n,c] and dscale[n,c] = (sum_gv - bias[c]*sum_g) / scale[n,c].
 *   mode 3  dx = g(upfirdn2d(x)),  g(.) = . * act'(.) * gain masked by the clamp, evaluated at yref = the forward output of the
 *           ACTIVATION that fed the (transposed) FIR -- yref has the shape of the result; sum_g[n,c] += sum(dx) (the bias gradient).
 *           The backward pass of "bias_act, then the FIR in front of a strided convolution" (DiscriminatorBlock conv0 -> conv1,
 *           networks.py:343-344 via conv2d_resample.py:113-126) in one pass instead of upfirdn2d + bias_act(grad 1).  scale / bias unused.
 *   mode 4  y = upfirdn2d(x) + yref,  yref shaped like y: a gradient added to the one that already arrived from the same tensor's other
 *           consumer (2x up-sampling geometry = the gradient of the 2x down-sampling FIR of the residual block's skip branch,
 *           networks.py:343; the other summand is conv0's data gradient).  act / scale / bias / sums unused.
 * act: 1 linear or 3 lrelu (bias_act.py:23-33 indices).  scale / bias may be NULL (1 / 0).  Supported for the
 * lane-exchange kernel's FIR geometries only (modes 1, 3: up=down=1, pad0 1; mode 2: up=down=1, pad0 2; mode 4: up=2, pad0 2; 4x4 filter,
 * dense NCHW); anything else returns SGV_ERR_UNSUPPORTED and the caller composes the three ops. */
typedef struct sgv_fir_epilogue {
    int32_t mode;
    const float* scale; /* [n*c] fp32 or NULL */
    const float* bias;  /* [c] fp32 or NULL */
    const void* yref;   /* mode 2: [n,c,in_h,in_w]; modes 3, 4: [n,c,out_h,out_w]; dtype/layout of x */
    float* sum_g;       /* modes 2, 3: [n*c], zero-initialised by the caller, accumulated with atomics */
    float* sum_gv;      /* mode 2: [n*c], likewise (mode 3: may be NULL) */
    int32_t act;
    float alpha, gain, clamp; /* clamp < 0: none */
} sgv_fir_epilogue;

int sgv_upfirdn2d_fused(const sgv_upfirdn2d_params* p, const sgv_fir_epilogue* e, int dtype, void* stream);

/* Which kernel sgv_upfirdn2d would run for p: 0 = generic gather kernel, 1 = register-window row walker
 * (contiguous NCHW, up/down in {1,2}, filter <= 4x4), 2 = lane-exchange kernel (the 2x up / 2x down hot-path geometries; the FIR
 * geometries when SGV_UFD_TILE=0), 3 = LDS-tile kernel (up = down = 1, filter <= 4x4, pad 0..3, output width % 4 in {0, 1}),
 * 4 = 2x down-sampling LDS-tile kernel (down = 2, 8 <= output width <= 128), 5 = 2x up-sampling LDS-tile kernel.
 * For tests/benchmarks. */
int sgv_upfirdn2d_kernel_kind(const sgv_upfirdn2d_params* p, int dtype);

/* ---------------------------------------------------------------------------------------
 * bias_act: y = clamp(act(x + b) * gain) and its first/second derivative forms.
 * Field meaning is that of bias_act.h:12-31.  act is the reference's cuda_idx (1..9:
 * linear, relu, lrelu, tanh, sigmoid, elu, selu, softplus, swish; bias_act.py:23-33).
 * grad: 0 forward, 1 first derivative (x holds dy), 2 second derivative (bias_act.cu:51-142).
 * NULL pointers mean "absent" (the reference passes an empty tensor).
 * clamp < 0 disables clamping.  All tensors share one dense layout of size_x elements.
 */
typedef struct sgv_bias_act_params {
    const void* x;
    const void* b;    /* [size_b] or NULL */
    const void* xref; /* or NULL */
    const void* yref; /* or NULL */
    const void* dy;   /* or NULL */
    void* y;
    int32_t grad;
    int32_t act;
    float alpha, gain, clamp;
    int32_t size_x;
    int32_t size_b;
    int32_t step_b; /* x.stride(dim) in elements (bias_act.cpp:73) */
} sgv_bias_act_params;

int sgv_bias_act(const sgv_bias_act_params* p, int dtype, void* stream);
/* grad = 1 form that also accumulates the bias gradient: db[slot][c] += partial sums of the result over channel c (fp32 atomics,
 * spread over db_slots (a power of two) copies to avoid same-address contention; the caller zero-initialises db[db_slots][size_b]
 * and adds the slots up).  Replaces `dx.sum([0, 2, 3])` of bias_act.py:185; needs step_b % (16 / sizeof(T)) == 0. */
int sgv_bias_act_db(const sgv_bias_act_params* p, float* db, int db_slots, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Weight (de)modulation of modulated_conv2d (networks.py:57-62) without the [N,O,I,kh,kw]
 * temporary:  d[n,o] = rsqrt( sum_i s[n,i]^2 * wsq[o,i] + eps ),  wsq[o,i] = sum_k W[o,i,k]^2.
 * All fp32.
 */
/* wsq[o*I + i] = sum_{k<kk} w[(o*I + i)*kk + k]^2 ; w contiguous [O, I, kk] */
int sgv_weight_sqsum(const float* w, float* wsq, int32_t oc, int32_t ic, int32_t kk, void* stream);
/* dcoefs[n*O + o] = rsqrt(sum_i (styles[n*I+i])^2 * wsq[o*I+i] + eps) */
int sgv_demod_coefs(const float* styles, const float* wsq, float* dcoefs, int32_t n, int32_t oc,
                    int32_t ic, float eps, void* stream);
/* First-order gradients of sgv_demod_coefs in one launch: with g = -grad_d * dcoefs^3 ([n, oc]), grad_w[o,i,k] = weight[o,i,k] * sum_n g[n,o] styles[n,i]^2 and
 * grad_s[n,i] = styles[n,i] * sum_o g[n,o] wsq[o,i]  (the backward of networks.py:59-61 without w[N,O,I,k,k]); grad_w or grad_s may be NULL.  kk = kh * kw. */
int sgv_demod_coefs_backward(const float* grad_d, const float* dcoefs, const float* styles, const float* wsq, const float* weight, float* grad_w, float* grad_s,
                             int32_t n, int32_t oc, int32_t ic, int32_t kk, void* stream);
/* y[n,c,hw] = x[n,c,hw] * s[n*C + c] (+ optional per-[n,hw] noise), contiguous NCHW,
 * x/y of dtype `dtype`, s fp32.  Used for x*styles and x*dcoefs (networks.py:66,70-71). */
/* out[p] += sum_i a[p,i] * b[p,i] over `planes` planes of hw elements (fp32 accumulate, atomics; the caller zero-initialises out):
 * the styles gradient of x * s[n,c] (`networks.py:66,70`) without materialising a * b. */
int sgv_plane_dot(const void* a, const void* b, float* out, int32_t planes, int32_t hw, int dtype, void* stream);
int sgv_scale_channels(const void* x, const float* s, void* y, int32_t n, int32_t c, int32_t hw,
                       int dtype, void* stream);
/* Backward helpers of the fused convolution layer (sgv_conv3x3_fused), fp32, dense [planes = n*c, hw]:
 * sgv_act_grad_scale: out = dz * d[plane] with dz = the bias_act gradient (grad 1, act 1 linear / 3 lrelu) of dy evaluated from the saved
 *   output y; sums (NULL or [2][planes], zero-initialised by the caller) += per-plane sum of dz and of dz * pre-activation
 *   (-> bias and demodulation-coefficient gradients).  d may be NULL (factor 1).
 * sgv_scale_dot: out = a * s[plane], dot[plane] += sum a * b  (input gradient and styles gradient of x * styles in one pass). */
int sgv_act_grad_scale(const float* dy, const float* y, const float* d, float* out, float* sums, int32_t planes, int32_t hw, int32_t act, float alpha,
                       float gain, float clamp, void* stream);
int sgv_scale_dot(const float* a, const float* b, const float* s, float* out, float* dot, int32_t planes, int32_t hw, void* stream);
/* The same two passes on fp32 / fp16 / bf16 tensors (dy, y, out resp. a, b, out in `dtype`; d, s, the sums and the dot products fp32; arithmetic in fp32,
 * one rounding on the store): the backward of the fused layer inside the mixed-precision blocks (networks.py:227,461 `num_fp16_res`). */
int sgv_act_grad_scale_t(const void* dy, const void* y, const float* d, void* out, float* sums, int32_t planes, int32_t hw, int32_t act, float alpha,
                         float gain, float clamp, int dtype, void* stream);
int sgv_scale_dot_t(const void* a, const void* b, const float* s, void* out, float* dot, int32_t planes, int32_t hw, int dtype, void* stream);
/* The same with `addend` (dtype / layout of out, or NULL) summed into the store: out = a * s[plane] + addend -- the gradient that reached the tensor from its OTHER
 * consumer (a synthesis block's output feeds the next block's up-sampling layer and its own ToRGB, networks.py:239-262), added here instead of by a pass of its own. */
int sgv_scale_dot_add_t(const void* a, const void* b, const float* s, const void* addend, void* out, float* dot, int32_t planes, int32_t hw, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * 1x1 convolutions with <= 4 channels on one side, as HBM streams (ToRGB Cin->3, fromRGB 3->C; NCHW, fp32 accumulate):
 *   kind 0 (many -> few):  y[n,f,p] = sum_m w[n,f,m] * x[n,m,p]     w: [n or 1][c_few][c_many] fp32
 *   kind 1 (few -> many):  y[n,m,p] = sum_f w[n,m,f] * x[n,f,p]     w: [n or 1][c_many][c_few] fp32
 * w_stride_n = elements between the weight sets of consecutive samples (0: shared).  hw = H*W must be a multiple of 4.
 * sgv_pointwise_outer is the weight-gradient reduction  out[n,f,m] += sum_p a[n,f,p] * b[n,m,p]  (fp32, atomics;
 * the caller zero-initialises `out`).
 */
typedef struct sgv_pointwise_params {
    const void* x;
    const float* w;
    void* y;
    int32_t n, c_many, c_few, hw;
    int64_t w_stride_n;
    int32_t kind;
} sgv_pointwise_params;

int sgv_pointwise_small(const sgv_pointwise_params* p, int dtype, void* stream);
/* kind 1 (few -> many) with the layer tail applied before the store: y = clamp(act(y + bias[m]) * gain), the same operations in the same
 * order as sgv_bias_act (bit-identical to the two-pass composition in fp32) -- the discriminator's fromRGB layer, layers.py Conv2dLayer.forward.
 * bias [c_many] fp32 or NULL; act 1 linear / 3 lrelu; clamp < 0: none; fp32 tensors only. */
int sgv_pointwise_act(const sgv_pointwise_params* p, const float* bias, int32_t act, float alpha, float gain, float clamp, int dtype, void* stream);
int sgv_pointwise_outer(const void* a_few, const void* b_many, float* out, int32_t n, int32_t c_few, int32_t c_many, int32_t hw,
                        int dtype, void* stream);
/* The backward pass of sgv_pointwise_act without materialising the activation gradient dz = bias_act'(dy; y) (grad 1 from the saved output,
 * linear / lrelu, gain, clamp mask):
 *   sgv_pointwise_small_gradin : kind 0 with x = dy turned into dz on its way in            -> the input gradient  dx[n,f,p] = sum_m w[f,m] dz[n,m,p]
 *   sgv_pointwise_outer_act    : out[n,f,m] += sum_p a[n,f,p] dz[n,m,p]; with ones_row the last of the c_few planes is the constant 1 (a holds
 *                                c_few - 1 planes), so that row c_few - 1 of out is the bias gradient    -> weight and bias gradients in one pass
 * fp32 tensors, 16-byte aligned. */
int sgv_pointwise_small_gradin(const sgv_pointwise_params* p, const void* yref, int32_t act, float alpha, float gain, float clamp, int dtype, void* stream);
int sgv_pointwise_outer_act(const void* a_few, const void* dy_many, const void* yref, float* out, int32_t n, int32_t c_few, int32_t c_many, int32_t hw,
                            int32_t ones_row, int32_t act, float alpha, float gain, float clamp, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Weight gradient of a 3x3 / stride 1 / pad 1 convolution on dense NCHW tensors:
 *   dw[o,i,ky,kx] = sum_{n,y,x} dy[n,o,y,x] * x[n,i,y+ky-1,x+kx-1]          dw: [c_out, c_in, 3, 3] fp32 (overwritten)
 * terms = 3: fp32 emulated on the bf16 matrix pipe with hi/lo splitting (bf16x3, ~2^-17 relative per product, fp32
 * accumulation); terms = 1: plain bf16 products.  sgv_conv3x3_wrw_supported() tells whether a shape/dtype is served
 * (fp32, channels % 64 == 0, and either w % 32 == 0 with h <= 32 or h % 32 == 0, or w in {16, 8} with h <= 32: 2 / 4 samples then share a
 * 32-pixel row step, any n); everything else stays with the vendor library.  dw is accumulated with atomics (summation order, and with it the
 * last bit, varies from run to run).
 */
typedef struct sgv_conv_wrw_params {
    const void* dy; /* [n, c_out, h, w] */
    const void* x;  /* [n, c_in, h, w] */
    float* dw;
    int32_t n, c_out, c_in, h, w;
    int32_t terms;          /* 1, 3, or 4 (block-scaled fp16 split: fp32-grade, see sgv_absmax) */
    const float* dy_amax;   /* terms = 4: device pointers to ONE fp32 each, upper bounds of max |dy| and max |x| */
    const float* x_amax;
    const float* x_amax2;   /* terms = 4, sgv_conv3x3_wrw_scaled only, REQUIRED there: a bound of |x_scale| (the operand is x * x_scale; a bound of x alone overflows fp16) */
} sgv_conv_wrw_params;

int sgv_conv3x3_wrw(const sgv_conv_wrw_params* p, int dtype, void* stream);
int sgv_conv3x3_wrw_supported(int32_t n, int32_t c_out, int32_t c_in, int32_t h, int32_t w, int dtype);
/* the same with x[n,i,:,:] * x_scale[n,i] as the input operand (x_scale fp32 [n, c_in]): the weight gradient of a modulated layer
 * (networks.py:66 `x * styles`) without materialising the scaled input again in the backward pass */
int sgv_conv3x3_wrw_scaled(const sgv_conv_wrw_params* p, const float* x_scale, int dtype, void* stream);
/* Stride-2 member (weight gradient of sgv_conv3x3_s2, either mode): dy = the SMALL tensor [n, c_out, h, w], x = the BIG one
 * [n, c_in, 2h+1, 2w+1];  dw[s,b,ky,kx] = sum_{n,Y,X} small[n,s,Y,X] * big[n,b,2Y+ky,2X+kx]  as [c_out, c_in, 3, 3].  Same shape rule as
 * the stride-1 member, on the small grid (w % 32 == 0, or w in {16, 8} with packed samples). */
int sgv_conv3x3_wrw_s2(const sgv_conv_wrw_params* p, int dtype, void* stream);
int sgv_conv3x3_wrw_s2_supported(int32_t n, int32_t c_small, int32_t c_big, int32_t h, int32_t w, int dtype);

/* ---------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution on dense NCHW tensors, forward and data gradient:
 *   mode 0:  y[n,m,Y,X] = sum_{k,ky,kx} weight[m][k][ky][kx]     * x[n,k,Y+ky-1,X+kx-1]     weight: [c_out, c_in, 3, 3]
 *   mode 1:  y[n,m,Y,X] = sum_{k,ky,kx} weight[k][m][2-ky][2-kx] * x[n,k,Y+ky-1,X+kx-1]     weight: [c_in, c_out, 3, 3]
 *            (= conv_transpose2d(x, weight, stride 1, padding 1): the gradient w.r.t. the input of the mode-0 layer)
 * Shapes: c_in % 16 == 0, c_out % 64 == 0, and either w % 32 == 0 && h % 16 == 0 or whole 16x16 / 8x8 images (n % 2 resp. n % 8 == 0); on the
 * w % 32 == 0 && h % 16 == 0 images also c_out % 32 == 0 (a half-full last 64-row tile; sgv_conv3x3_workspace_bytes counts whole tiles).
 * fp32 tensors, arithmetic as sgv_conv3x3_wrw (terms = 3: bf16x3 fp32 emulation; 1: bf16 products).  `workspace` is
 * sgv_conv3x3_workspace_bytes() of device scratch for the re-laid-out weights (owned by the caller, written per call).
 *
 * `dtype` = SGV_F16 / SGV_BF16 (every member of the 3x3 family: sgv_conv3x3[_fused], sgv_conv3x3_s2[_fused], sgv_conv3x3_wrw[_scaled],
 * sgv_conv3x3_wrw_s2): x and y (dy and x; act_out) are 16-bit tensors, the weight, the scales, the bias and the weight gradient stay fp32 -- the
 * mixed-precision blocks of the reference (`num_fp16_res`, networks.py:227,461; `weight.to(x.dtype)` networks.py:67) with fp32 master weights
 * handed in as they are.  Needs terms = 1: every value is ONE bf16 operand of the matrix pipe (a bf16 as it is, an fp16 or fp32 weight rounded to
 * nearest even), fp32 accumulate, one rounding on the store.  Shapes: the big-image forms only (w % 32 == 0; stride 2 also w in {16, 8} on the small
 * grid; strided form c_out % 128 == 0); no `accumulate` (the residual add is an fp32 atomic).  Tolerance against fp32: 1e-2 of the result's scale
 * (tests/test_conv_lowp_gpu.py; exact on bf16-representable integer data).
 */
typedef struct sgv_conv3x3_params {
    const void* x;        /* [n, c_in, h, w] */
    const float* weight;  /* fp32, contiguous */
    void* y;              /* [n, c_out, h, w] */
    void* workspace;
    int64_t workspace_bytes;
    int32_t n, c_in, c_out, h, w;
    int32_t mode;
    int32_t terms;         /* 1 bf16 products, 3 bf16 split (bf16x3), 4 block-scaled fp16 split (fp32-grade; see below) */
    const float* x_amax;   /* terms = 4: device pointer to ONE fp32 >= max |x| (sgv_absmax writes one); ignored otherwise */
    const float* x_amax2;  /* terms = 4: a second factor of the bound -- REQUIRED with x_scale (sgv_conv3x3_fused: a bound of |x_scale|), ignored without; */
    const float* w_amax;   /* terms = 4, optional: a bound of max |weight| the caller already has (a weight is bounded once per optimiser step, not once per
                              launch); NULL: the library runs its own pass over the weight */
} sgv_conv3x3_params;

/*
 * Stride-2 members (sgv_conv3x3_s2; h, w name the SMALL H x W grid, the big tensor is (2h+1) x (2w+1), no padding):
 *   mode 0:  y[n,m,Y,X]        = sum_{k,ky,kx} weight[m][k][ky][kx] * x[n,k,2Y+ky,2X+kx]      x big -> y small (conv2d, stride 2)
 *   mode 2:  y[n,m,2Y+ky,2X+kx] += weight[k][m][ky][kx] * x[n,k,Y,X]                          x small -> y big (conv_transpose2d, stride 2)
 * These are the convolutions either side of the FIR in the reference's down- and up-sampling layers
 * (conv2d_resample.py:113-137) and each other's data gradients.  Shapes: sgv_conv3x3_s2_supported_mode (c_in % 16, c_out % 64 -- mode 2 also
 * c_out % 32 --, w % 32, h % 8 on the small grid, or w in {16, 8} with packed samples).
 */
int sgv_conv3x3(const sgv_conv3x3_params* p, int dtype, void* stream);
int sgv_conv3x3_s2(const sgv_conv3x3_params* p, int dtype, void* stream);

/*
 * terms = 4 -- fp32-grade products on the 16-bit matrix pipe (every member of the 3x3 family; csrc/sgv_split.h).  The reference's config-3
 * arithmetic is strict fp32 (`torch.backends.cudnn.allow_tf32 = False`, src/training/training_loop.py:129,141-142).  Each operand tensor is scaled by
 * a power of two that puts its largest magnitude just below fp16's maximum, split into two fp16 terms (22 significant bits) and multiplied with
 * three MFMAs into fp32 accumulators, like terms = 3; the result is scaled back exactly.  The scale comes from a BOUND of the tensor's
 * magnitude that the caller passes as a device pointer (`x_amax`, `dy_amax`): any value >= max |x|, tight within ~2^10.  sgv_absmax computes
 * max |x| itself in one streaming pass (out[0] = max |x|; accumulate != 0: max with the value already there); weights are bounded by the library.
 * A bound smaller than the data overflows fp16 (inf / NaN in the result).
 */
int sgv_absmax(const void* x, int64_t numel, int dtype, float* out, int32_t accumulate, void* stream);
/* The same bound as a free by-product of the kernel that WRITES the tensor: sgv_amax_sink(out) arms a one-shot side output for the NEXT sgv_* call of
 * this thread.  `out` is a block of 1 + SGV_AMAX_SINK_SLOTS floats that must hold zeros when that call's kernels run: the producer's waves fold their
 * maxima into the slots behind out[0] (spread, so that ~10^6 short-lived waves do not queue on one address), a one-workgroup kernel launched by the
 * same call folds the slots into out[0].  If that call's kernel supports it (the LDS-tile forms of sgv_upfirdn2d / sgv_upfirdn2d_fused modes 1 and 3, sgv_act_grad_scale[_t],
 * sgv_scale_channels, sgv_pointwise_act -- all on fp32 tensors) out[0] = max |output| after it, and sgv_amax_sink_consumed() returns 1; any other
 * call leaves `out` untouched, disarms the sink and sgv_amax_sink_consumed() returns 0 (the caller then runs sgv_absmax on the result).  Thread-local;
 * nothing else in the library is stateful. */
#define SGV_AMAX_SINK_SLOTS 4096
int sgv_amax_sink(float* out);
int sgv_amax_sink_consumed(void);

/*
 * sgv_conv3x3 with the element-wise steps that surround the convolution of a stride-1 SynthesisLayer / Conv2dLayer folded in
 * (modulated_conv2d training path networks.py:65-74 + `bias_act` networks.py:141-143, layers.py Conv2dLayer.forward):
 *
 *     y[n,m] = clamp( act( out_scale[n,m] * conv3x3( x[n,k] * x_scale[n,k] , weight )[n,m] + bias[m] ) * gain )
 *
 * x_scale (styles) multiplies the activations on their way into the matrix-core operand tiles; out_scale (demodulation
 * coefficients), bias, activation (1 = linear, 3 = lrelu with `alpha`), gain and clamp (< 0: none) are applied to the fp32
 * accumulators before the only store.  Any of the three pointers may be NULL (factor 1 / no bias).  Needs gain > 0 and
 * 0 <= alpha <= 1.  Result equals the three-pass composition up to fused-multiply-add rounding of the epilogue.
 * Served for the shapes of sgv_conv3x3_fused_supported() (the big-image kernel: W % 32 == 0, H % 16 == 0).
 */
typedef struct sgv_conv3x3_epilogue {
    const float* x_scale;    /* [n, c_in]  or NULL */
    const float* out_scale;  /* [n, c_out] or NULL */
    const float* bias;       /* [c_out]    or NULL */
    int32_t act;             /* 1 linear, 3 lrelu */
    float alpha, gain, clamp;
    int32_t accumulate;      /* != 0: y += result (one fp32 add per element) -- e.g. a data gradient summed into the gradient that already
                                arrived from the layer input's other consumer (act 1, gain 1, no bias / scales: the bare convolution) */
} sgv_conv3x3_epilogue;
int sgv_conv3x3_fused(const sgv_conv3x3_params* p, const sgv_conv3x3_epilogue* e, int dtype, void* stream);
int sgv_conv3x3_fused_supported(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int dtype);
int sgv_conv3x3_s2_supported(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int dtype);
/* Strided 3x3 layer with its tail fused into the accumulator store (layers.py Conv2dLayer.forward after conv2d_resample's FIR; the residual
 * add of DiscriminatorBlock.forward, networks.py:343-345):
 *     a = clamp(act(conv3x3_s2(x, weight) + bias[m]) * gain),   y = a   or, with accumulate != 0,   y += a
 * act 1 linear / 3 lrelu(alpha), gain > 0, 0 <= alpha <= 1, clamp < 0: none.  `accumulate` is the reference's in-place `y.add_(x)`: p->y holds
 * the skip branch's result on entry and each element receives one fp32 add.  bias / act_out may be NULL; act_out receives a (what the activation
 * gradient needs once the sum hides it).  Shapes served: sgv_conv3x3_s2_fused_supported (mode 0, c_out % 128 == 0, H % 8 == 0). */
typedef struct sgv_conv3x3_s2_epilogue {
    const float* bias;       /* [c_out], 16-byte aligned, or NULL */
    float* act_out;          /* [n, c_out, h, w] or NULL */
    int32_t act;
    float alpha, gain, clamp;
    int32_t accumulate;
} sgv_conv3x3_s2_epilogue;
int sgv_conv3x3_s2_fused(const sgv_conv3x3_params* p, const sgv_conv3x3_s2_epilogue* e, int dtype, void* stream);
int sgv_conv3x3_s2_fused_supported(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int dtype);
/* as sgv_conv3x3_s2_supported, per mode: the transposed form (mode 2) also serves W = 16 / 8 with n % (32 / W) == 0, H % 4 == 0 */
int sgv_conv3x3_s2_supported_mode(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int32_t mode, int dtype);
int64_t sgv_conv3x3_s2_workspace_bytes(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int32_t mode);
int sgv_conv3x3_supported(int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int dtype);
int64_t sgv_conv3x3_workspace_bytes(int32_t c_in, int32_t c_out);

/* ---------------------------------------------------------------------------------------
 * Bilinear resampling under one affine map per sample (ADA geometric execution: `affine_grid(theta, size, align_corners=False)` +
 * `grid_sample(bilinear, zeros)` of augment.py:297-300, without materialising the grid), fp32 NCHW:
 *   adjoint = 0:  dst[n,c,Y,X] = bilinear sample of src[n,c] at theta[n] @ (xn, yn, 1)     src [n,c,h,w] -> dst [n,c,ho,wo]
 *   adjoint = 1:  the transposed map (gradient w.r.t. the image): src [n,c,ho,wo] -> dst [n,c,h,w], dst zero-initialised by the caller
 * theta is [n, 2, 3] in affine_grid's normalised coordinates. */
int sgv_affine_resample(const float* src, float* dst, const float* theta, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ho, int32_t wo,
                        int32_t adjoint, void* stream);

/* The whole geometric execution of AugmentPipe.forward (src/training/augment.py:270-300) as one kernel, forward direction, fp32 NCHW:
 *   y = upfirdn2d.downsample2d( grid_sample( upfirdn2d.upsample2d( pad(x, [mx0, mx1, my0, my1], 'reflect'), Hz, up=2 ), affine_grid(theta) ), Hz, down=2,
 *                               padding=-6, flip_filter=True )                                      x, y [n, c, h, w]
 * with the 12-tap `Hz_geom` filter (`filter12`: HOST pointer, the taps travel as launch arguments) and theta [n, 2, 3] (DEVICE) exactly what
 * sgv_affine_resample would get for the resampling step: the map from the (2 (h + 6)) x (2 (w + 6)) resampled image to the up-sampled padded image
 * (2 (h + my0 + my1)) x (2 (w + mx0 + mx1)), both in affine_grid's normalised coordinates.  The padded, up-sampled and resampled images are never
 * written.  Margins in [0, size - 1] (the reference clamps there, augment.py:281-282).  Any affine map: samples whose 16 x 16 tile footprints do not fit the
 * staging buffers are served in 8 x 8 / 4 x 4 sub-tiles by a second kernel (two launches, each sample in exactly one), extreme zoom-outs by a direct form. */
int sgv_ada_geometric(const float* x, float* y, const float* theta, const float* filter12, int32_t n, int32_t c, int32_t h, int32_t w,
                      int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, void* stream);

/* The adjoint of sgv_ada_geometric (same arguments, same coefficient and tap arithmetic): dx = A^T dy for the linear map y = A x of that call; dy, dx [n, c, h, w].
 * With sgv_ada_geometric it serves every order of derivative w.r.t. the image (the backward of the reference's block: the backward of grid_sample_gradfix.py:45-83,
 * of upfirdn2d.py:249-260 twice and of the reflect pad).  dx is written, not accumulated.  Any affine map: samples with a singular / non-finite map or a zoom-in
 * beyond ~3.4 are served by an atomics kernel behind the main one (two launches; the second returns at once for every other sample).  The bilinear transpose adds
 * into LDS words in no fixed order (the reference's own backward, ATen's grid_sampler_2d_backward, uses global atomics): results repeat to rounding, not bitwise. */
int sgv_ada_geometric_adjoint(const float* dy, float* dx, const float* theta, const float* filter12, int32_t n, int32_t c, int32_t h, int32_t w,
                              int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, void* stream);

/* ---------------------------------------------------------------------------------------
 * AlignedTimeEncoder element-wise tail (motion.py:201-212), fp32:
 *   raw(tau) = freqs[j]*periods[r,j]*tau + phases[r,j]*phase_scales[j]
 *   pos(tau) = [sin raw(tau) | cos raw(tau)]                       (2*nf wide)
 *   out[r,:] = pos(t) - lerp(pos(t_left), pos(t_right), alpha) + lerp(al[r,:], ar[r,:], alpha)
 * periods already include the tanh()+1 of motion.py:196.  Accurate sinf/cosf (arguments reach
 * ~1e3 rad).  rows = batch*frames.
 */
typedef struct sgv_time_encode_params {
    const float* periods; /* [rows, nf] */
    const float* phases;  /* [rows, nf] */
    const float* al;      /* [rows, 2*nf] aligners from the left code  */
    const float* ar;      /* [rows, 2*nf] aligners from the right code */
    const float* freqs;        /* [nf] */
    const float* phase_scales; /* [nf] */
    const float* t;       /* [rows] */
    const float* t_left;  /* [rows] */
    const float* t_right; /* [rows] */
    const float* alpha;   /* [rows] */
    float* out;           /* [rows, 2*nf] */
    int32_t rows, nf;
} sgv_time_encode_params;

int sgv_time_encode(const sgv_time_encode_params* p, void* stream);

/* ---------------------------------------------------------------------------------------
 * Dense (batched) GEMM on the matrix cores, fp32 in / fp32 accumulate on
 * v_mfma_f32_32x32x2_f32 (exact fp32, same rounding as an fmaf chain over k):
 *     C[b][M,N] = A[b][M,K] * op(B[b]) (+ bias),   all row-major, ld* in elements.
 *   trans_b = 1: B is stored [N,K]  ->  FullyConnectedLayer `x @ w.t()` (layers.py:133-137):
 *                A = activations [rows, in], B = weight [out, in], bias per column (bias_mode 1).
 *   trans_b = 0: B is stored [K,N]  ->  a 1x1 convolution on NCHW (conv2d_resample.py:40-54):
 *                A = weight [Cout, Cin] shared by the batch (stride_a = 0), B = x[n] [Cin, H*W],
 *                C = y[n] [Cout, H*W], bias per row (bias_mode 2).
 */
typedef struct sgv_gemm_params {
    const float* a;
    const float* b;
    const float* bias; /* or NULL */
    float* c;
    int32_t m, n, k;
    int64_t lda, ldb, ldc;
    int32_t trans_b;
    int32_t batch;
    int64_t stride_a, stride_b, stride_c; /* elements between consecutive batch entries */
    int32_t bias_mode;                    /* 0 none, 1 bias[N] per column, 2 bias[M] per row */
    int32_t k_split;                      /* > 1: K is cut into k_split equal slices, slice s of batch entry b is written to c + (b * k_split + s) * stride_c
                                             (the caller sums them; no bias) -- long-K products with few output tiles, e.g. the 1x1 weight gradients */
    const float* residual;                /* or NULL: C = A * op(B) (+ bias) + residual, residual laid out like C (ldc, stride_c) -- the sum of the
                                             discriminator block's two branches (networks.py:343-345) formed in the skip convolution's store */
    int32_t exact_fp32;                   /* 0: products as 2-way bf16 splits on the bf16 matrix pipe (3 MFMAs per product, fp32 accumulate, 4.4e-6 relative error:
                                             the arithmetic of the 3x3 family) where the shape allows (n % 128 == 0, k % 32 == 0, 16-byte aligned rows);
                                             1: always v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain over k).  SGV_GEMM_TERMS=0 forces 1 for the process;
                                             2: the split members with block-scaled fp16 operands (fp32-grade, see sgv_absmax; needs a_amax / b_amax) where the
                                                shape allows, the exact pipe elsewhere */
    const float* a_amax;                  /* exact_fp32 = 2: device pointers to ONE fp32 each, upper bounds of max |A| / max |B| over the whole tensors */
    const float* b_amax;
} sgv_gemm_params;

int sgv_gemm_f32(const sgv_gemm_params* p, void* stream);

/* ---------------------------------------------------------------------------------------
 * Small-M dense layer with FullyConnectedLayer's element-wise steps folded in (fp32, exact-fp32 MFMA; csrc/fc.hip).
 *   C(m,n) = epi( sum_k A(m,k) * B(k,n) ),   A(m,k) = a[m*a_stride_m + k*a_stride_k],  B(k,n) = b[k*b_stride_k + n*b_stride_n]
 *   epi(v) = act(v * weight_gain + bias[n] * bias_gain) * gain        (act / gain only if epilogue_act)
 * replaces `torch.addmm` / `matmul` + `bias_act` of layers.py:108-138.  Optional:
 *   normalize_a : rows of A are scaled by rsqrt(mean_k A^2 + 1e-8) (normalize_2nd_moment of the mapping input, layers.py:22-25)
 *   a_ref       : same indexing as a; A(m,k) becomes the bias_act gradient (grad 1) of a w.r.t. the saved OUTPUT a_ref:
 *                 ((a_ref > 0) ? a : a * alpha) * gain  for act = 3 (lrelu), a * gain for act = 1  -- the backward forms read dy and y directly
 *   a_rowsum    : [m] <- bias_gain * sum_k A(m,k)  (the bias gradient when A = dz^T)
 * The three forms of a dense layer y = act(x W^T wg + b bg) * gain with x [M,K], W [N,K]:
 *   forward     A = x (K,1)        B = W^T (1,K)      C = y [M,N]
 *   data grad   A = dy (N,1; a_ref = y)   B = W (K,1)        C = dx [M,K]    weight_gain = wg
 *   weight grad A = dy^T (1,N; a_ref = y) B = x (K,1)        C = dW [N,K]    weight_gain = wg, a_rowsum = db
 */
typedef struct sgv_fc_params {
    const float* a; int64_t a_stride_m, a_stride_k;
    const float* a_ref;
    const float* b; int64_t b_stride_k, b_stride_n;
    float* c; int64_t c_stride_m, c_stride_n;
    const float* bias;
    float* a_rowsum;
    int32_t m, n, k;
    int32_t normalize_a;
    int32_t act;            /* 1 linear, 3 lrelu */
    float alpha, gain;
    float weight_gain, bias_gain;
    int32_t epilogue_act;
    int32_t batch;          /* >= 1 independent problems: pointers advance by the *_stride_batch elements (0 = shared operand); batch == 0 means 1 */
    int64_t a_stride_batch, b_stride_batch, c_stride_batch;
    int32_t accumulate;     /* C += epi(...) instead of C = epi(...) */
} sgv_fc_params;
int sgv_fc(const sgv_fc_params* p, void* stream);
/* `count` independent problems (HOST array) in one launch per 24 problems: the style affines of a synthesis pass (src/training/networks.py:116,128,153,160 --
 * `self.affine(w)` of every SynthesisLayer / ToRGBLayer) forward, their data gradients and their weight gradients.  The problems of one call may differ in every
 * pointer, size, row stride, weight gain and activation gain; they share the operand contiguity (a_stride_k == 1, b_stride_k == 1 or not), act, alpha, epilogue_act,
 * bias_gain, c_stride_n and accumulate, and none is batched or normalising (SGV_ERR_UNSUPPORTED otherwise). */
int sgv_fc_grouped(const sgv_fc_params* problems, int32_t count, void* stream);

/* ---------------------------------------------------------------------------------------
 * In-place torch.nan_to_num(x, nan, posinf, neginf) over a LIST of fp32 tensors in one launch per 96 tensors: the per-parameter gradient
 * sanitising loop of src/training/training_loop.py:384-386.  `tensors` / `numels` are HOST arrays of `count` device pointers / element
 * counts; the table is passed in the kernel arguments, nothing is copied or allocated. */
int sgv_multi_nan_to_num_f32(float* const* tensors, const int64_t* numels, int32_t count, float nan, float posinf, float neginf, void* stream);

/* ---------------------------------------------------------------------------------------
 * dst[i] = src[i] * scales[i] over a LIST of dense fp32 tensors in one launch per 64 tensors: the equalised learning-rate scaling of a module's
 * convolution weights and biases (src/training/layers.py:184-185: `w = self.weight * (self.weight_gain * self.lr_multiplier)`,
 * `b = self.bias * self.lr_multiplier`, evaluated per layer and again per gradient by the reference).  `src` / `dst` / `numels` / `scales` are HOST
 * arrays of `count` entries; the table travels in the kernel arguments (capture-safe).  src[i] and dst[i] may be the same tensor. */
int sgv_multi_scale_f32(const float* const* src, float* const* dst, const int64_t* numels, const float* scales, int32_t count, void* stream);

/* ---------------------------------------------------------------------------------------
 * Per-launch timing (bench.py roofline leg).  When enabled, every sgv_* launch is bracketed by
 * two HIP events on its own stream; sgv_prof_collect synchronises those events and reports, per
 * kernel family, launch count, summed milliseconds and summed algorithmic bytes
 * (upfirdn2d: (numel(x)+numel(y))*sizeof(T); bias_act: all streams read + written).
 */
enum sgv_kernel_family {
    SGV_K_UPFIRDN2D_ROWS = 0,
    SGV_K_UPFIRDN2D_GENERIC = 1,
    SGV_K_BIAS_ACT = 2,
    SGV_K_MODULATE = 3,
    SGV_K_TIME_ENCODE = 4,
    SGV_K_GEMM = 5,
    SGV_K_UPFIRDN2D_LANES = 6,
    SGV_K_POINTWISE = 7,
    SGV_K_CONV_WRW = 8,
    SGV_K_CONV3X3 = 9,          /* every 3x3 convolution launch except the >= 32 pixel stride-1 kernel */
    SGV_K_CONV3X3_S1 = 10,      /* conv3x3_ws_kernel (stride 1, forward and data gradient, images >= 32 pixels): the step's dominant kernel */
    SGV_K_FC = 11,              /* sgv_fc (dense layers; the tiled GEMM keeps SGV_K_GEMM) */
    SGV_K_ABSMAX = 12,          /* sgv_absmax: the magnitude-bound passes of the block-scaled fp16 split (timed, not counted by sgv_launch_count) */
    SGV_K_COUNT = 13
};
typedef struct sgv_prof_entry {
    int64_t launches;
    double ms;
    double bytes; /* algorithmic bytes */
    double flops; /* algorithmic flops (GEMM), else 0 */
} sgv_prof_entry;

int sgv_prof_enable(int32_t max_records); /* starts a new record list; grows the event pool and the device table of timestamp pairs (16 bytes per record) when needed */
int sgv_prof_disable(void);
/* Records again WITHOUT resetting the pool (sgv_prof_enable starts a new one).  A launch bracketed while its stream is being CAPTURED is timed by two one-thread
 * kernels that store the device's constant-rate clock -- nodes of the graph like the launch itself: every replay rewrites the pair, and the collect calls then
 * read that launch's duration inside the LAST replay (1.03; event records inside a capture cannot be read back under the HIP runtime PyTorch bundles). */
int sgv_prof_resume(void);
/* Bracket only the launches of the families whose bit (1 << enum sgv_kernel_family) is set; 0 = all (the default).  bench.py times the dominant kernels inside
 * its timed region with two families enabled (two events per launch of 554 launches per iteration cost 1.4 % of the step) and the full table in a pass of its own. */
int sgv_prof_families(uint64_t mask);
int sgv_prof_collect(sgv_prof_entry* out /* [SGV_K_COUNT] */); /* syncs events, resets pool */
/* The same, launch by launch (in launch order) instead of summed per family: fills at most `max_records` entries, returns the number of recorded
 * launches (which may exceed max_records), or a negative SGV_ERR_* code.  Resets the pool like sgv_prof_collect. */
typedef struct sgv_prof_record {
    int32_t family;  /* enum sgv_kernel_family */
    int32_t variant; /* index into sgv_variant_name(): the kernel variant the call took (the first one it noted), or -1 */
    float ms;        /* the whole call: auxiliary launches (weight preparation, edge strips, memset) included */
    float reserved;
    double bytes, flops;
} sgv_prof_record;
int sgv_prof_collect_records(sgv_prof_record* out, int32_t max_records);

/* Total number of kernel launches issued through this library since load (all threads). */
int64_t sgv_launch_count(void);

/* Which kernel served a call: one counter per kernel variant (the members of a family that a shape / switch selects between), bumped
 * at launch time.  sgv_variant_name(i) is NULL for i outside [0, number of variants); sgv_variant_count(i) is -1 there. */
int64_t sgv_variant_count(int32_t variant);
const char* sgv_variant_name(int32_t variant);

int sgv_version(void);
const char* sgv_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SGV_OPS_H */
