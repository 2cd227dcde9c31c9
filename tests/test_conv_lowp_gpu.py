"""16-bit (bf16 / fp16) tensor I/O of the stride-1 3x3 convolution and weight-gradient kernels (csrc/sgv_io16.h) through the C ABI.

The mixed-precision blocks of the reference (`num_fp16_res`, src/training/networks.py:227,461) hand fp16 activations and `weight.to(x.dtype)`
to cuDNN.  Here the kernels read the 16-bit activations and the fp32 master weight, turn every value into ONE bf16 operand of the matrix
pipe, accumulate in fp32 and write 16-bit outputs (fp32 weight gradients).  What that arithmetic must satisfy, and what is asserted:

  * integer data (|x| <= 3, |w| <= 2: exact as bf16, exact products, exact fp32 sums): the output equals the float64 oracle rounded once
    to the tensor format -- bit-exact; weight gradients (fp32) equal the oracle exactly.  Pins indexing for both element sizes.
  * random data, oracle evaluated on the SAME operands the kernel multiplies (the 16-bit activations as they are for bf16, rounded to bf16
    for fp16; the weight rounded to bf16): what is left is fp32 summation order + the one output rounding -- |err| <= 2^-8 |ref| + 1e-5*scale
    for bf16 outputs, 2^-11 for fp16 outputs; weight gradients < 1e-5 of scale.
  * random data against the float64 oracle on the unrounded fp32 weight: the STATED 16-bit tolerance, 1e-2 of the output's scale (measured
    ~3e-3: 2^-9 per rounded operand over a 576..4608-term sum, plus the output rounding).
"""
import numpy as np
import pytest
import torch

import oracle
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import conv2d_gradfix

pytestmark = pytest.mark.gpu
DEV = 'cuda'
S1 = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
S1T = (True, (1, 1), (1, 1), (0, 0), (1, 1), 1)
DTYPES = [torch.bfloat16, torch.float16]
ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def _bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _conv(x, w, transposed):
    cfg = S1T if transposed else S1
    assert conv2d_gradfix._native_conv_ok(x, w, cfg), 'this 16-bit shape is not served by the hand-written kernel'
    before = custom_ops.kernel_variant_counts().get('conv_lowp', 0)
    y = conv2d_gradfix._native_conv(x, w, cfg)
    assert custom_ops.kernel_variant_counts().get('conv_lowp', 0) == before + 1
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
    same_operands = _ref_conv(_bf16(x.float()), _bf16(w), transposed)
    scale = np.abs(same_operands).max()
    err = np.abs(y - same_operands)
    assert (err <= ULP[dtype] * np.abs(same_operands) + 1e-5 * scale).all(), f'worst {err.max() / scale:.2e} of scale beyond summation order + one output rounding'
    full = _ref_conv(x, w, transposed)
    tol = np.abs(y - full).max() / np.abs(full).max()
    print(f'[{dtype} {ci}->{co} {h}x{wd}{" T" if transposed else ""}] error vs float64 on the fp32 weight: {tol:.2e} of scale')
    assert tol < 1e-2


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n,o,i,h,wd', [(2, 64, 64, 32, 32), (1, 64, 128, 64, 64), (2, 128, 64, 64, 96), (1, 192, 64, 96, 32)])
def test_conv3x3_s1_weight_gradient_16bit_tensors(dtype, n, o, i, h, wd):
    g = torch.Generator().manual_seed(n * 100 + o + i + h)
    shape = (o, i, 3, 3)

    def run(dy, x):
        assert conv2d_gradfix._native_wrw_ok(dy, x, S1, shape), 'this 16-bit shape is not served by the hand-written kernel'
        before = custom_ops.kernel_variant_counts().get('wrw_lowp', 0)
        dw = conv2d_gradfix._native_wrw(dy, x, S1, shape)
        assert custom_ops.kernel_variant_counts().get('wrw_lowp', 0) == before + 1
        assert dw.dtype == torch.float32
        return dw.double().cpu().numpy()

    dyi = torch.randint(-3, 4, [n, o, h, wd], generator=g).to(DEV).to(dtype)
    xi = torch.randint(-3, 4, [n, i, h, wd], generator=g).to(DEV).to(dtype)
    assert np.array_equal(run(dyi, xi), oracle.conv3x3_weight_grad(dyi.double().cpu().numpy(), xi.double().cpu().numpy()))
    dy = torch.randn([n, o, h, wd], generator=g).to(DEV).to(dtype)
    x = (torch.randn([n, i, h, wd], generator=g) * 1.5 + 0.25).to(DEV).to(dtype)
    got = run(dy, x)
    same = oracle.conv3x3_weight_grad(_bf16(dy.float()).double().cpu().numpy(), _bf16(x.float()).double().cpu().numpy())
    assert np.abs(got - same).max() / np.abs(same).max() < 1e-5
    full = oracle.conv3x3_weight_grad(dy.double().cpu().numpy(), x.double().cpu().numpy())
    tol = np.abs(got - full).max() / np.abs(full).max()
    print(f'[{dtype} dw {o}x{i} {h}x{wd}] error vs float64 on the unrounded tensors: {tol:.2e} of scale')
    assert tol < (1e-5 if dtype == torch.bfloat16 else 1e-2)   # bf16 tensors ARE the operands; fp16 tensors lose 3 mantissa bits on the way in


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
    dw = conv2d_gradfix._native_wrw(dyi, xi, S1, (c, c, 3, 3))
    dw32 = conv2d_gradfix._native_wrw(dyi.float(), xi.float(), S1, (c, c, 3, 3))
    assert torch.equal(dw, dw32)


@pytest.mark.parametrize('dtype', DTYPES)
def test_autograd_with_fp32_master_weight(dtype):
    """conv2d_gradfix.conv2d(x16, w32): y in the tensor format, dx in the tensor format, dw in fp32 -- and the double-backward pieces run."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn([2, 64, 32, 32], generator=g).to(DEV).to(dtype).requires_grad_(True)
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    assert conv2d_gradfix.cast_weight(w, x) is w
    y = conv2d_gradfix.conv2d(x, w, padding=1)
    assert y.dtype == dtype
    dy = torch.randn(y.shape, generator=g).to(DEV).to(dtype)
    dx, dw = torch.autograd.grad(y, [x, w], dy, create_graph=True)
    assert dx.dtype == dtype and dw.dtype == torch.float32
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, padding=1)
    dxr, dwr = torch.autograd.grad(yr, [xr, wr], dy.double(), create_graph=True)
    for got, ref in ((y, yr), (dx, dxr), (dw, dwr)):
        assert ((got.double() - ref).abs().max() / ref.abs().max()).item() < 1e-2
    (gx2,) = torch.autograd.grad((dx.float() ** 2).sum() + (dw ** 2).sum(), [x], allow_unused=True)   # R1-style second order through both nodes
    (gx2r,) = torch.autograd.grad((dxr ** 2).sum() + (dwr ** 2).sum(), [xr])
    assert gx2.dtype == dtype
    assert ((gx2.double() - gx2r).abs().max() / gx2r.abs().max()).item() < 2e-2
