"""16-bit (bf16 / fp16) tensor I/O of the 3x3 convolution and weight-gradient kernels -- stride 1, stride 2, transposed stride 2, and the forms that
pack 2 / 4 narrow samples into a tile row (csrc/sgv_io16.h) -- through the C ABI.

The mixed-precision blocks of the reference (`num_fp16_res`, src/training/networks.py:227,461) hand fp16 activations and `weight.to(x.dtype)`
to cuDNN.  Here the kernels read the 16-bit activations and the fp32 master weight, turn every value into ONE 16-bit operand of the matrix
pipe -- a bf16 for bf16 tensors, an fp16 for fp16 tensors (round 5: v_mfma_f32_32x32x16_f16 on the activations AS THEY ARE and on the weight rounded
to fp16, exactly what the reference multiplies; rounds 1-4 rounded fp16 values to bf16 and lost 3 bits) -- accumulate in fp32 and write 16-bit
outputs (fp32 weight gradients).  What that arithmetic must satisfy, and what is asserted:

  * integer data (|x| <= 3, |w| <= 2: exact in both formats, exact products, exact fp32 sums): the output equals the float64 oracle rounded once
    to the tensor format -- bit-exact; weight gradients (fp32) equal the oracle exactly.  Pins indexing for both element sizes.
  * random data, oracle evaluated on the SAME operands the kernel multiplies (the 16-bit activations as they are, the weight rounded to the
    tensor format): what is left is fp32 summation order + the one output rounding -- |err| <= 2^-8 |ref| + 1e-5*scale
    for bf16 outputs, 2^-11 for fp16 outputs; weight gradients < 1e-5 of scale in both formats (their operands are the tensors).
  * random data against the float64 oracle on the unrounded fp32 weight: the STATED tolerances -- bf16: 1e-2 of the output's scale (measured
    ~3e-3: 2^-9 per rounded operand over a 576..4608-term sum, plus the output rounding); fp16: 2e-3 of scale and 1e-3 rel-L2 (measured ~4e-4).
"""
import numpy as np
import pytest
import torch

import oracle
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import conv2d_gradfix
from util import dispatch_assert

pytestmark = pytest.mark.gpu
DEV = 'cuda'
S1 = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
S1T = (True, (1, 1), (1, 1), (0, 0), (1, 1), 1)
DTYPES = [torch.bfloat16, torch.float16]
ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


STATED = {torch.bfloat16: 1e-2, torch.float16: 2e-3}       # max |err| / max |ref| against float64 on the unrounded weight
STATED_L2 = {torch.bfloat16: 1e-2, torch.float16: 1e-3}    # rel-L2


def _op(t, dtype):
    """The operand the kernel multiplies for a tensor of format `dtype`: the value rounded to that format (activations: unchanged)."""
    return t.to(dtype).to(t.dtype)


def _conv(x, w, transposed):
    cfg = S1T if transposed else S1
    dispatch_assert(conv2d_gradfix._native_conv_ok(x, w, cfg), 'this 16-bit shape is not served by the hand-written kernel')
    before = custom_ops.kernel_variant_counts().get('conv_lowp', 0)
    y = conv2d_gradfix._native_conv(x, w, cfg)
    dispatch_assert(custom_ops.kernel_variant_counts().get('conv_lowp', 0) == before + 1)
    assert y.dtype == x.dtype
    return y


def _ref_conv(x, w, transposed):
    return oracle.conv3x3(x.double().cpu().numpy(), w.double().cpu().numpy(), stride=1, transposed=transposed)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('transposed', [False, True])
@pytest.mark.parametrize('n,ci,co,h,wd', [(2, 64, 64, 32, 32), (1, 128, 64, 48, 64), (3, 64, 128, 32, 96), (1, 16, 64, 64, 32)])
def test_conv3x3_s1_16bit_tensors(dtype, transposed, n, ci, co, h, wd):
    g = torch.Generator().manual_seed(n + ci + co + h + (7 if transposed else 0))
    wshape = [ci, co, 3, 3] if transposed else [co, ci, 3, 3]
    # integer data: one rounding of the exact result
    xi = torch.randint(-3, 4, [n, ci, h, wd], generator=g).to(DEV).to(dtype)
    wi = torch.randint(-2, 3, wshape, generator=g).float().to(DEV)
    yi = _conv(xi, wi, transposed)
    want = torch.from_numpy(_ref_conv(xi, wi, transposed)).to(dtype)
    assert torch.equal(yi.cpu(), want), f'{int((yi.cpu() != want).sum())} of {want.numel()} elements differ from the once-rounded exact result'
    # random data
    x = (torch.randn([n, ci, h, wd], generator=g) + 0.25).to(DEV).to(dtype)
    w = (torch.randn(wshape, generator=g) / (3 * ci ** 0.5)).to(DEV)
    y = _conv(x, w, transposed).double().cpu().numpy()
    same_operands = _ref_conv(_op(x.float(), dtype), _op(w, dtype), transposed)
    scale = np.abs(same_operands).max()
    err = np.abs(y - same_operands)
    assert (err <= ULP[dtype] * np.abs(same_operands) + 1e-5 * scale).all(), f'worst {err.max() / scale:.2e} of scale beyond summation order + one output rounding'
    full = _ref_conv(x, w, transposed)
    tol = np.abs(y - full).max() / np.abs(full).max()
    l2 = np.linalg.norm(y - full) / np.linalg.norm(full)
    print(f'[{dtype} {ci}->{co} {h}x{wd}{" T" if transposed else ""}] error vs float64 on the fp32 weight: {tol:.2e} of scale, rel-L2 {l2:.2e}')
    assert tol < STATED[dtype] and l2 < STATED_L2[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n,o,i,h,wd', [(2, 64, 64, 32, 32), (1, 64, 128, 64, 64), (2, 128, 64, 64, 96), (1, 192, 64, 96, 32)])
def test_conv3x3_s1_weight_gradient_16bit_tensors(dtype, n, o, i, h, wd):
    g = torch.Generator().manual_seed(n * 100 + o + i + h)
    shape = (o, i, 3, 3)

    def run(dy, x):
        dispatch_assert(conv2d_gradfix._native_wrw_ok(dy, x, S1, shape), 'this 16-bit shape is not served by the hand-written kernel')
        before = custom_ops.kernel_variant_counts().get('wrw_lowp', 0)
        dw = conv2d_gradfix._native_wrw(dy, x, S1, shape)
        dispatch_assert(custom_ops.kernel_variant_counts().get('wrw_lowp', 0) == before + 1)
        assert dw.dtype == torch.float32
        return dw.double().cpu().numpy()

    dyi = torch.randint(-3, 4, [n, o, h, wd], generator=g).to(DEV).to(dtype)
    xi = torch.randint(-3, 4, [n, i, h, wd], generator=g).to(DEV).to(dtype)
    assert np.array_equal(run(dyi, xi), oracle.conv3x3_weight_grad(dyi.double().cpu().numpy(), xi.double().cpu().numpy()))
    dy = torch.randn([n, o, h, wd], generator=g).to(DEV).to(dtype)
    x = (torch.randn([n, i, h, wd], generator=g) * 1.5 + 0.25).to(DEV).to(dtype)
    got = run(dy, x)
    same = oracle.conv3x3_weight_grad(dy.double().cpu().numpy(), x.double().cpu().numpy())
    assert np.abs(got - same).max() / np.abs(same).max() < 1e-5
    full = oracle.conv3x3_weight_grad(dy.double().cpu().numpy(), x.double().cpu().numpy())
    tol = np.abs(got - full).max() / np.abs(full).max()
    print(f'[{dtype} dw {o}x{i} {h}x{wd}] error vs float64 on the unrounded tensors: {tol:.2e} of scale')
    assert tol < 1e-5   # the tensors ARE the operands in both formats


@pytest.mark.parametrize('dtype', DTYPES)
def test_scaled_weight_gradient_16bit_tensors(dtype):
    """The modulated layers' weight gradient takes x * s[n, c] (fp32 scale applied before the operand is rounded)."""
    if not conv2d_gradfix.wrw_input_scale:
        pytest.skip('scaled weight gradients switched off')
    g = torch.Generator().manual_seed(5)
    n, o, i, h, wd = 3, 64, 128, 32, 64
    dy = torch.randint(-3, 4, [n, o, h, wd], generator=g).to(DEV).to(dtype)
    x = torch.randint(-3, 4, [n, i, h, wd], generator=g).to(DEV).to(dtype)
    s = torch.randint(1, 4, [n, i], generator=g).float().to(DEV)
    dispatch_assert(conv2d_gradfix._native_wrw_kind(dy, x, S1, (o, i, 3, 3)) == 's1', 'this 16-bit shape is not served by the hand-written kernel')
    dw = conv2d_gradfix._native_wrw(dy, x, S1, (o, i, 3, 3), x_scale=s)
    want = oracle.conv3x3_weight_grad(dy.double().cpu().numpy(), (x.double() * s.double()[:, :, None, None]).cpu().numpy())
    assert np.array_equal(dw.double().cpu().numpy(), want)


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv3x3_s1_16bit_at_the_benchmark_shape(dtype):
    """96 frames of 64 -> 64 channels at 256x256 (6144 tiles on 256 persistent workgroups): integer data, first / middle / last frame slabs
    exact, every plane sum of the fp32 twin reproduced after the output rounding."""
    g = torch.Generator(device=DEV).manual_seed(11)
    n, c, r = 96, 64, 256
    xi = torch.randint(-3, 4, [n, c, r, r], generator=g, device=DEV).to(dtype)
    wi = torch.randint(-2, 3, [c, c, 3, 3], generator=g, device=DEV).float()
    y = _conv(xi, wi, False)
    y32 = conv2d_gradfix._native_conv(xi.float(), wi, S1)        # the fp32 member: exact on this data (test_conv_bench_shapes_gpu.py)
    assert torch.equal(y, y32.to(dtype))
    for f, r0 in ((0, 0), (n // 2 + 1, 121), (n - 1, r - 6)):
        lo, hi = max(r0 - 1, 0), min(r0 + 7, r)
        ref = oracle.conv3x3(xi[f:f + 1, :, lo:hi].double().cpu().numpy(), wi.double().cpu().numpy())[:, :, r0 - lo:r0 - lo + 6]
        assert torch.equal(y[f:f + 1, :, r0:r0 + 6].cpu(), torch.from_numpy(ref).to(dtype))
    dyi = torch.randint(-2, 3, [n, c, r, r], generator=g, device=DEV).to(dtype)
    dispatch_assert(conv2d_gradfix._native_wrw_kind(dyi, xi, S1, (c, c, 3, 3)) == 's1', 'this 16-bit shape is not served by the hand-written kernel')
    dw = conv2d_gradfix._native_wrw(dyi, xi, S1, (c, c, 3, 3))
    dw32 = conv2d_gradfix._native_wrw(dyi.float(), xi.float(), S1, (c, c, 3, 3))
    assert torch.equal(dw, dw32)


@pytest.mark.parametrize('dtype', DTYPES)
def test_autograd_with_fp32_master_weight(dtype):
    """conv2d_gradfix.conv2d(x16, w32): y in the tensor format, dx in the tensor format, dw in fp32 -- and the double-backward pieces run."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn([2, 64, 32, 32], generator=g).to(DEV).to(dtype).requires_grad_(True)
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    dispatch_assert(conv2d_gradfix.cast_weight(w, x) is w, 'fp32 master weights reach the 16-bit kernels as they are')
    y = conv2d_gradfix.conv2d(x, w, padding=1)
    assert y.dtype == dtype
    dy = torch.randn(y.shape, generator=g).to(DEV).to(dtype)
    dx, dw = torch.autograd.grad(y, [x, w], dy, create_graph=True)
    assert dx.dtype == dtype and dw.dtype == torch.float32
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, padding=1)
    dxr, dwr = torch.autograd.grad(yr, [xr, wr], dy.double(), create_graph=True)
    for got, ref in ((y, yr), (dx, dxr), (dw, dwr)):
        assert ((got.double() - ref).abs().max() / ref.abs().max()).item() < STATED[dtype]
    (gx2,) = torch.autograd.grad((dx.float() ** 2).sum() + (dw ** 2).sum(), [x], allow_unused=True)   # R1-style second order through both nodes
    (gx2r,) = torch.autograd.grad((dxr ** 2).sum() + (dwr ** 2).sum(), [xr])
    assert gx2.dtype == dtype
    assert ((gx2.double() - gx2r).abs().max() / gx2r.abs().max()).item() < 2e-2


# ------------------------------------------------------------------------------------------------------------------------------------------
# stride 2: the strided form (x (2H+1)x(2W+1) -> y HxW, the tap-pair kernel), the transposed form (x HxW -> y (2H+1)x(2W+1), producer / consumer kernel
# + the MFMA edge strips) and their weight gradient; W = 16 / 8 on the small grid: 2 / 4 samples per tile row
S2 = (False, (2, 2), (0, 0), (0, 0), (1, 1), 1)
S2T = (True, (2, 2), (0, 0), (0, 0), (1, 1), 1)


def _conv_s2(x, w, transposed):
    cfg = S2T if transposed else S2
    dispatch_assert(conv2d_gradfix._native_conv_kind(x, w, cfg) == 's2', 'this 16-bit stride-2 shape is not served by the hand-written kernel')
    name = 'convT_lowp' if transposed else 'conv_s2_lowp'
    before = custom_ops.kernel_variant_counts().get(name, 0)
    y = conv2d_gradfix._native_conv(x, w, cfg)
    dispatch_assert(custom_ops.kernel_variant_counts().get(name, 0) == before + 1)
    assert y.dtype == x.dtype
    return y


def _ref_s2(x, w, transposed):
    return oracle.conv3x3(x.double().cpu().numpy(), w.double().cpu().numpy(), stride=2, transposed=transposed)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('transposed', [False, True])
@pytest.mark.parametrize('n,ci,co,hs,ws', [(2, 64, 128, 8, 32), (1, 128, 128, 16, 64), (3, 16, 128, 8, 32), (4, 64, 128, 16, 16), (8, 32, 128, 8, 8), (2, 64, 256, 24, 32)])
def test_conv3x3_s2_16bit_tensors(dtype, transposed, n, ci, co, hs, ws):
    """hs x ws: the small grid.  Strided: x [n, ci, 2hs+1, 2ws+1] -> y [n, co, hs, ws]; transposed: x [n, ci, hs, ws] -> y [n, co, 2hs+1, 2ws+1]."""
    g = torch.Generator().manual_seed(n + ci + co + hs + ws + (7 if transposed else 0))
    xshape = [n, ci, hs, ws] if transposed else [n, ci, 2 * hs + 1, 2 * ws + 1]
    wshape = [ci, co, 3, 3] if transposed else [co, ci, 3, 3]
    xi = torch.randint(-3, 4, xshape, generator=g).to(DEV).to(dtype)
    wi = torch.randint(-2, 3, wshape, generator=g).float().to(DEV)
    yi = _conv_s2(xi, wi, transposed)
    want = torch.from_numpy(_ref_s2(xi, wi, transposed)).to(dtype)
    assert yi.shape == want.shape
    bad = yi.cpu() != want
    assert not bad.any(), f'{int(bad.sum())} of {want.numel()} elements differ from the once-rounded exact result (last row {int(bad[:, :, -1].sum())}, last column {int(bad[:, :, :, -1].sum())})'
    x = (torch.randn(xshape, generator=g) + 0.25).to(DEV).to(dtype)
    w = (torch.randn(wshape, generator=g) / (3 * ci ** 0.5)).to(DEV)
    y = _conv_s2(x, w, transposed).double().cpu().numpy()
    same_operands = _ref_s2(_op(x.float(), dtype), _op(w, dtype), transposed)
    scale = np.abs(same_operands).max()
    err = np.abs(y - same_operands)
    assert (err <= ULP[dtype] * np.abs(same_operands) + 1e-5 * scale).all(), f'worst {err.max() / scale:.2e} of scale beyond summation order + one output rounding'
    full = _ref_s2(x, w, transposed)
    tol = np.abs(y - full).max() / np.abs(full).max()
    l2 = np.linalg.norm(y - full) / np.linalg.norm(full)
    print(f'[{dtype} s2 {ci}->{co} small grid {hs}x{ws}{" T" if transposed else ""}] error vs float64 on the fp32 weight: {tol:.2e} of scale, rel-L2 {l2:.2e}')
    assert tol < STATED[dtype] and l2 < STATED_L2[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('transposed', [False, True])
@pytest.mark.parametrize('n,cs,cb,hs,ws', [(2, 64, 64, 8, 32), (1, 128, 64, 32, 64), (4, 64, 128, 16, 16), (5, 64, 64, 8, 8), (2, 64, 64, 64, 32)])
def test_conv3x3_s2_weight_gradient_16bit_tensors(dtype, transposed, n, cs, cb, hs, ws):
    """cs / cb: channels of the small (hs x ws) / big ((2hs+1) x (2ws+1)) tensor; the weight is [cs, cb, 3, 3] for the strided layer (dy small, x big) and
    for the transposed one (x small, dy big)."""
    g = torch.Generator().manual_seed(n * 10 + cs + cb + hs + ws)
    cfg = S2T if transposed else S2
    shape = (cs, cb, 3, 3)

    def run(small, big):
        dy, x = (big, small) if transposed else (small, big)
        dispatch_assert(conv2d_gradfix._native_wrw_kind(dy, x, cfg, shape) == 's2', 'this 16-bit stride-2 shape is not served by the hand-written kernel')
        before = custom_ops.kernel_variant_counts().get('wrw_s2_lowp', 0)
        dw = conv2d_gradfix._native_wrw(dy, x, cfg, shape)
        dispatch_assert(custom_ops.kernel_variant_counts().get('wrw_s2_lowp', 0) == before + 1)
        assert dw.dtype == torch.float32
        return dw.double().cpu().numpy()

    def ref(small, big):   # the strided layer's formula; the transposed layer has the same one with the roles of x and dy swapped
        return oracle.conv3x3_weight_grad(small.double().cpu().numpy(), big.double().cpu().numpy(), stride=2)

    si = torch.randint(-3, 4, [n, cs, hs, ws], generator=g).to(DEV).to(dtype)
    bi = torch.randint(-3, 4, [n, cb, 2 * hs + 1, 2 * ws + 1], generator=g).to(DEV).to(dtype)
    assert np.array_equal(run(si, bi), ref(si, bi))
    sm = torch.randn([n, cs, hs, ws], generator=g).to(DEV).to(dtype)
    bg = (torch.randn([n, cb, 2 * hs + 1, 2 * ws + 1], generator=g) * 1.5 + 0.25).to(DEV).to(dtype)
    got = run(sm, bg)
    same = ref(sm, bg)
    assert np.abs(got - same).max() / np.abs(same).max() < 1e-5
    full = ref(sm, bg)
    tol = np.abs(got - full).max() / np.abs(full).max()
    print(f'[{dtype} dw s2 {cs}x{cb} small grid {hs}x{ws}] error vs float64 on the unrounded tensors: {tol:.2e} of scale')
    assert tol < 1e-5


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv3x3_s2_16bit_at_the_benchmark_shape(dtype):
    """96 frames, 64 <-> 128 channels between 257x257 and 128x128: the 16-bit kernels reproduce their fp32 twins (exact on integer data, pinned to the
    oracle in test_conv_bench_shapes_gpu.py) after the one output rounding, in both directions, and the weight gradient exactly."""
    g = torch.Generator(device=DEV).manual_seed(12)
    n = 96
    big = torch.randint(-3, 4, [n, 64, 257, 257], generator=g, device=DEV).to(dtype)
    small = torch.randint(-3, 4, [n, 128, 128, 128], generator=g, device=DEV).to(dtype)
    w = torch.randint(-2, 3, [128, 64, 3, 3], generator=g, device=DEV).float()
    y = _conv_s2(big, w, False)
    assert torch.equal(y, conv2d_gradfix._native_conv(big.float(), w, S2).to(dtype))
    yt = _conv_s2(small, w, True)
    assert torch.equal(yt, conv2d_gradfix._native_conv(small.float(), w, S2T).to(dtype))
    dispatch_assert(conv2d_gradfix._native_wrw_kind(small, big, S2, (128, 64, 3, 3)) == 's2', 'this 16-bit stride-2 shape is not served by the hand-written kernel')
    dw = conv2d_gradfix._native_wrw(small, big, S2, (128, 64, 3, 3))
    assert torch.equal(dw, conv2d_gradfix._native_wrw(small.float(), big.float(), S2, (128, 64, 3, 3)))


# ------------------------------------------------------------------------------------------------------------------------------------------
# whole layers on 16-bit activations (ops/fused_conv_act.py, ops/fused_down_act.py): the fused kernels against the float64 composition
def _rel_l2(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return ((got - ref).norm() / ref.norm()).item()


def _lrelu_layer_f64(x, w, s, d, b, stride, positive=None):
    """float64 layer.  `positive` (bool, the output's shape): use THIS sign pattern for the leaky relu instead of the pre-activation's own -- the gradient
    reference: a 16-bit evaluation flips the sign of the few pre-activations that lie within its rounding error of zero, and the backward pass of any
    implementation (the reference's fp16 one included: bias_act.py:185 masks with the stored output) then follows its own forward pattern."""
    x = x * s[:, :, None, None] if s is not None else x
    y = torch.nn.functional.conv2d(x, w, padding=1 if stride == 1 else 0, stride=stride)
    y = y * d[:, :, None, None] if d is not None else y
    y = y + b[None, :, None, None]
    positive = y > 0 if positive is None else positive
    return torch.where(positive, y, 0.2 * y) * np.sqrt(2)


def _check_layer(name, y, grads, ins, dy, s6d6, stride, dtype=torch.bfloat16):
    """Forward against the float64 layer; gradients against the float64 layer differentiated on the kernel's own sign pattern; the patterns themselves
    differ in < 0.5 % of the elements.  Stated tolerances (rel-L2): bf16 1e-2, fp16 1e-3 (STATED_L2)."""
    ins64 = [t.detach().double().requires_grad_(True) for t in ins]
    x6, w6, b6 = ins64[:3]
    s6, d6 = (ins64[3], ins64[4]) if s6d6 else (None, None)
    y6 = _lrelu_layer_f64(x6, w6, s6, d6, b6, stride)
    flips = ((y > 0) != (y6 > 0)).float().mean().item()
    ym = _lrelu_layer_f64(x6, w6, s6, d6, b6, stride, positive=(y > 0))
    grads6 = torch.autograd.grad(ym, ins64, dy.double())
    errs = [_rel_l2(y, y6)] + [_rel_l2(a, r_) for a, r_ in zip(grads, grads6)]
    print(f'[{name}] rel-L2 of y and of the gradients (x, w, b, ...):', ' '.join(f'{e:.1e}' for e in errs), f'| sign flips {flips:.1e}')
    assert max(errs) < STATED_L2[dtype] and flips < 5e-3
    return y6


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('modulated', [False, True])
def test_fused_stride1_layer_on_16bit_activations(dtype, modulated):
    from stylegan_v_amd.torch_utils.ops import fused_conv_act
    g = torch.Generator().manual_seed(21)
    n, ci, co, r = 4, 64, 128, 64
    x = torch.randn([n, ci, r, r], generator=g).to(DEV).to(dtype).requires_grad_(True)
    w = (torch.randn([co, ci, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    b = (torch.randn([co], generator=g) * 0.1).to(DEV).requires_grad_(True)
    s = (torch.rand([n, ci], generator=g) + 0.5).to(DEV).requires_grad_(True) if modulated else None
    d = (torch.rand([n, co], generator=g) + 0.5).to(DEV).requires_grad_(True) if modulated else None
    ins = [t for t in (x, w, b, s, d) if t is not None]
    before = custom_ops.kernel_variant_counts()
    y = fused_conv_act.conv3x3_bias_act(x, w, styles=s, dcoefs=d, bias=b, act='lrelu')
    after = custom_ops.kernel_variant_counts()
    dispatch_assert(after.get('conv_s1_ws_fused', 0) == before.get('conv_s1_ws_fused', 0) + 1 and after.get('conv_lowp', 0) == before.get('conv_lowp', 0) + 1)
    assert y.dtype == dtype
    dy = torch.randn(y.shape, generator=g).to(DEV).to(dtype)
    grads = torch.autograd.grad(y, ins, dy)
    assert grads[0].dtype == dtype and all(t.dtype == torch.float32 for t in grads[1:])
    y6 = _check_layer(f'{dtype} fused s1 layer, modulated={modulated}', y, grads, ins, dy, modulated, 1, dtype)
    # and no further from float64 than the same layer evaluated op by op in the tensor format
    with fused_conv_act.composition_only():
        yc = fused_conv_act.conv3x3_bias_act(x, w, styles=s, dcoefs=d, bias=b, act='lrelu')
    assert _rel_l2(y, y6) <= 1.5 * _rel_l2(yc, y6) + 1e-4


@pytest.mark.parametrize('dtype', DTYPES)
def test_fused_downsampling_layer_on_16bit_activations(dtype):
    from stylegan_v_amd.torch_utils.ops import fused_down_act
    g = torch.Generator().manual_seed(22)
    n, ci, co, hs = 4, 64, 128, 32
    xb = torch.randn([n, ci, 2 * hs + 1, 2 * hs + 1], generator=g).to(DEV).to(dtype).requires_grad_(True)
    w = (torch.randn([co, ci, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    b = (torch.randn([co], generator=g) * 0.1).to(DEV).requires_grad_(True)
    before = custom_ops.kernel_variant_counts()
    y = fused_down_act.strided_conv3x3_bias_act(xb, w, bias=b, act='lrelu')
    after = custom_ops.kernel_variant_counts()
    dispatch_assert(after.get('conv_s2_pairs_fused', 0) == before.get('conv_s2_pairs_fused', 0) + 1 and after.get('conv_s2_lowp', 0) == before.get('conv_s2_lowp', 0) + 1)
    assert y.dtype == dtype
    dy = torch.randn(y.shape, generator=g).to(DEV).to(dtype)
    grads = torch.autograd.grad(y, [xb, w, b], dy)
    assert grads[0].dtype == dtype and grads[1].dtype == torch.float32
    _check_layer(f'{dtype} fused down layer', y, grads, [xb, w, b], dy, False, 2, dtype)
    after2 = custom_ops.kernel_variant_counts()
    dispatch_assert(after2.get('convT_lowp', 0) == after.get('convT_lowp', 0) + 1 and after2.get('wrw_s2_lowp', 0) == after.get('wrw_s2_lowp', 0) + 1)


@pytest.mark.parametrize('dtype', DTYPES)
def test_nan_activation_propagates_through_the_16bit_convolution(dtype):
    """A NaN activation must come out as NaN wherever its 3 x 3 footprint reaches, as in the reference's (and this repo's fp32) arithmetic -- the fp16 operand
    path saturated with v_med3_f32, which returns a FINITE value for a NaN input and hid the divergence."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn([1, 64, 32, 32], generator=g)
    x[0, 3, 10, 17] = float('nan')
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV)
    y = _conv(x.to(DEV).to(dtype), w, False).float().cpu()
    bad = torch.isnan(y[0])
    assert bad[:, 9:12, 16:19].all(), 'the NaN did not reach every output of its footprint'
    clean = bad.clone()
    clean[:, 9:12, 16:19] = False
    assert not clean.any(), 'NaN outside the footprint'
