"""The FUSED entry points of the headline step at the shapes the benchmark runs them (BASELINE config 3: N = 96 frames, FFS-256
channel ladder), through the C ABI, against oracle/oracle.py.

`tests/test_conv_bench_shapes_gpu.py` covers the plain convolution entry points at these sizes; more than half of the dominant
kernel's launches in the step are the fused instantiations (`conv3x3_ws_kernel<3, PRO, EPI = 1>`, the `accumulate` store,
`conv3x3_s2_pairs_kernel<..., EPI = 1>` with the residual, `sgv_upfirdn2d_fused` modes 1-3).  `tests/test_fused_conv_gpu.py` checks
those against the oracle at a handful of tiles -- every workgroup owns at most one.  Here every launch has 1,536-12,288 tiles on 256
persistent workgroups, so the epilogue vectors travelling with the chunk sets across tile boundaries are what is being tested.

Scheme (same as the plain tests):
  * random data: output slabs (first / middle / last frame; top / interior / bottom rows; all channels) against the float64 oracle
    COMPOSITION (oracle.conv3x3 -> * dcoefs + bias -> oracle.bias_act ...): relative error < 1e-5 of the slab's scale;
  * exactly representable data (small integers, power-of-two scales, lrelu slope 0.25, gain 2): the same slabs EXACT against the
    oracle composition, and the WHOLE output tensor exact against the epilogue formula applied (plain torch arithmetic, every step exact)
    to the plain entry point's output -- which test_conv_bench_shapes_gpu.py pins with its exact plane checksums on the same kind of data;
  * upfirdn2d_fused: bit-exact against the oracle composition on sampled planes and against the three-pass composition of this
    library's own ops on the whole tensor (fp32: same operations in the same order); plane sums within fp32 summation-order rounding.
"""
import numpy as np
import pytest
import torch

import oracle
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import conv2d_gradfix, fused_conv_act, fused_down_act, fused_fir_act, upfirdn2d
from util import dispatch_assert

pytestmark = pytest.mark.gpu
DEV = 'cuda'
N = 96
S1 = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
S1T = (True, (1, 1), (1, 1), (0, 0), (1, 1), 1)
S2 = (False, (2, 2), (0, 0), (0, 0), (1, 1), 1)
ALPHA, GAIN = 0.25, 2.0     # exactly representable activation constants for the integer runs


def _one_conv_launch(fn):
    custom_ops.prof_enable(16)
    out = fn()
    custom_ops.prof_disable()
    dispatch_assert(custom_ops.prof_collect()['conv3x3']['launches'] == 1, 'the fused layer must be ONE convolution-family launch')
    return out


def _lrelu_formula(acc, d, b, alpha, gain, clamp=None):
    """clamp(lrelu(acc * d + b) * gain) in plain fp32 torch arithmetic (exact on the integer data of these tests)."""
    v = acc
    if d is not None:
        v = v * d[:, :, None, None]
    if b is not None:
        v = v + b[None, :, None, None]
    v = torch.where(v > 0, v, v * alpha) * gain
    return v.clamp(-clamp, clamp) if clamp is not None else v


def _slabs(ho, h):
    rows = 6 if h >= 64 else min(ho, 8)
    return [(0, 0, rows), (N // 2 + 1, (ho - rows) // 2 + 1, (ho - rows) // 2 + 1 + rows), (N - 1, ho - rows, ho)]


def _s1_slab_oracle(x, w, s, d, b, n, r0, r1, alpha, gain, transposed=False):
    """float64 oracle composition for rows [r0, r1) of frame n of a stride-1 layer."""
    h = x.shape[2]
    lo, hi = max(r0 - 1, 0), min(r1 + 1, h)
    xs = x[n:n + 1, :, lo:hi].double().cpu()
    if s is not None:
        xs = xs * s[n:n + 1].double().cpu()[:, :, None, None]
    acc = torch.from_numpy(oracle.conv3x3(xs.numpy(), w.double().cpu().numpy(), stride=1, transposed=transposed))[:, :, r0 - lo:r0 - lo + (r1 - r0)].contiguous()   # oracle.bias_act reads dense memory
    if d is not None:
        acc = acc * d[n:n + 1].double().cpu()[:, :, None, None]
    return acc, oracle.bias_act(acc, b.double().cpu() if b is not None else None, act='lrelu', alpha=alpha, gain=gain)


# ---------------------------------------------------------------------------------------------------------------------------------------
# sgv_conv3x3_fused, forward: PRO = 0 / 1 x EPI = 1

@pytest.mark.parametrize('c,r', [(64, 256), (512, 32)])
@pytest.mark.parametrize('modulated', [False, True])
def test_fused_stride1_layer_at_benchmark_shapes(c, r, modulated):
    """SynthesisLayer conv1 (styles prologue + dcoefs / bias / lrelu in the store, networks.py:65-74,141-143) and DiscriminatorBlock conv0
    (bias / lrelu in the store, layers.py Conv2dLayer.forward) at N = 96."""
    g = torch.Generator(device=DEV).manual_seed(7 * c + r + modulated)
    x = torch.randn([N, c, r, r], generator=g, device=DEV) + 0.25
    w = torch.randn([c, c, 3, 3], generator=g, device=DEV) / (3 * c ** 0.5)
    s = (torch.randn([N, c], generator=g, device=DEV) * 0.3 + 1) if modulated else None
    d = (torch.rand([N, c], generator=g, device=DEV) + 0.5) if modulated else None
    b = torch.randn([c], generator=g, device=DEV) * 0.5
    alpha, gain = 0.2, 2 ** 0.5
    assert custom_ops.get_native().sgv_conv3x3_fused_supported(N, c, c, r, r, 0)
    y = _one_conv_launch(lambda: fused_conv_act._launch_fused(x, w, s, d, b, 3, alpha, gain, -1.0))
    worst = 0.0
    for n, r0, r1 in _slabs(r, r):
        _, ref = _s1_slab_oracle(x, w, s, d, b, n, r0, r1, alpha, gain)
        err = (y[n:n + 1, :, r0:r1].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, err)
        assert err < 1e-5, f'frame {n} rows {r0}:{r1}: relative error {err:.2e} vs the float64 oracle composition'

    # exactly representable data: integers, styles in {-1, 1, 2}, dcoefs in {0.5, 1, 2}, integer bias, slope 0.25, gain 2
    xi = torch.randint(-3, 4, x.shape, generator=g, device=DEV).float()
    wi = torch.randint(-2, 3, w.shape, generator=g, device=DEV).float()
    si = torch.tensor([-1.0, 1.0, 2.0], device=DEV)[torch.randint(0, 3, [N, c], generator=g, device=DEV)] if modulated else None
    di = torch.tensor([0.5, 1.0, 2.0], device=DEV)[torch.randint(0, 3, [N, c], generator=g, device=DEV)] if modulated else None
    bi = torch.randint(-40, 41, [c], generator=g, device=DEV).float()
    yi = _one_conv_launch(lambda: fused_conv_act._launch_fused(xi, wi, si, di, bi, 3, ALPHA, GAIN, -1.0))
    for n, r0, r1 in _slabs(r, r):
        _, refi = _s1_slab_oracle(xi, wi, si, di, bi, n, r0, r1, ALPHA, GAIN)
        assert torch.equal(yi[n:n + 1, :, r0:r1].double().cpu(), refi), f'frame {n} rows {r0}:{r1}: integer data is not exact'
    plain = conv2d_gradfix._native_conv(xi * si[:, :, None, None] if modulated else xi, wi, S1)    # pinned by the exact plane checksums of test_conv_bench_shapes_gpu.py
    want = _lrelu_formula(plain, di, bi, ALPHA, GAIN)
    assert torch.equal(yi, want), f'{int((yi != want).sum())} of {yi.numel()} elements of the whole fused output differ from epilogue(plain convolution)'
    # with a clamp (conv_clamp of the mixed-precision configs): the same, saturating
    yc = fused_conv_act._launch_fused(xi, wi, si, di, bi, 3, ALPHA, GAIN, 64.0)
    assert torch.equal(yc, _lrelu_formula(plain, di, bi, ALPHA, GAIN, clamp=64.0))
    print(f'[fused s1 {c}ch {r}^2 modulated={modulated}] worst slab error {worst:.2e}')


# ---------------------------------------------------------------------------------------------------------------------------------------
# sgv_conv3x3_fused, mode 1 + accumulate: the data gradient that adds into the gradient the skip branch produced

@pytest.mark.parametrize('c,r', [(64, 256), (512, 32)])
def test_fused_data_gradient_accumulates_at_benchmark_shapes(c, r):
    """_FusedConvActFirFn.backward: conv0's data gradient lands IN the skip branch's gradient (`accumulate` of sgv_conv3x3_epilogue)."""
    g = torch.Generator(device=DEV).manual_seed(11 * c + r)
    dz = torch.randn([N, c, r, r], generator=g, device=DEV)
    w = torch.randn([c, c, 3, 3], generator=g, device=DEV) / (3 * c ** 0.5)
    base = torch.randn([N, c, r, r], generator=g, device=DEV)
    acc = base.clone()
    out = _one_conv_launch(lambda: fused_conv_act._launch_fused(dz, w, None, None, None, 1, 0.0, 1.0, -1.0, mode=1, accumulate_into=acc))
    assert out.data_ptr() == acc.data_ptr()
    for n, r0, r1 in _slabs(r, r):
        conv, _ = _s1_slab_oracle(dz, w, None, None, None, n, r0, r1, 1.0, 1.0, transposed=True)
        ref = conv + base[n:n + 1, :, r0:r1].double().cpu()
        err = (acc[n:n + 1, :, r0:r1].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, f'frame {n} rows {r0}:{r1}: relative error {err:.2e}'
    dzi = torch.randint(-3, 4, dz.shape, generator=g, device=DEV).float()
    wi = torch.randint(-2, 3, w.shape, generator=g, device=DEV).float()
    basei = torch.randint(-100, 101, base.shape, generator=g, device=DEV).float()
    acci = basei.clone()
    fused_conv_act._launch_fused(dzi, wi, None, None, None, 1, 0.0, 1.0, -1.0, mode=1, accumulate_into=acci)
    want = conv2d_gradfix._native_conv(dzi, wi, S1T) + basei
    assert torch.equal(acci, want), f'{int((acci != want).sum())} elements of base + data gradient are not exact'


# ---------------------------------------------------------------------------------------------------------------------------------------
# sgv_conv3x3_s2_fused: strided convolution + bias + lrelu + gain (+ residual add of the discriminator block)

@pytest.mark.parametrize('n,cb,cs,hs', [(N, 64, 128, 128), (N, 256, 512, 32), (32, 512, 512, 8), (N, 512, 512, 16)])
@pytest.mark.parametrize('with_res', [True, False])
def test_fused_down_layer_at_benchmark_shapes(n, cb, cs, hs, with_res):
    """DiscriminatorBlock conv1 with its tail and `y.add_(x)` (networks.py:343-345): 64 -> 128 at 257 -> 128, 256 -> 512 at 65 -> 32 and the
    packed-sample forms (17 -> 8 of the concatenated-frames block: 32 videos; 33 -> 16)."""
    g = torch.Generator(device=DEV).manual_seed(cb + cs + hs + with_res)
    hb = 2 * hs + 1
    assert custom_ops.get_native().sgv_conv3x3_s2_fused_supported(n, cb, cs, hs, hs, 0)
    xb = torch.randn([n, cb, hb, hb], generator=g, device=DEV) + 0.25
    w = torch.randn([cs, cb, 3, 3], generator=g, device=DEV) / (3 * cb ** 0.5)
    b = torch.randn([cs], generator=g, device=DEV) * 0.5
    res0 = torch.randn([n, cs, hs, hs], generator=g, device=DEV) if with_res else None
    alpha, gain = 0.2, 1.0          # sqrt(2) lrelu gain x sqrt(0.5) residual gain
    y, a = _one_conv_launch(lambda: fused_down_act._launch(xb, w, b, res0.clone() if with_res else None, with_res, 3, alpha, gain, -1.0))
    rows = min(hs, 6)
    slabs = [(0, 0, rows), (n // 2 + 1, (hs - rows) // 2, (hs - rows) // 2 + rows), (n - 1, hs - rows, hs)]
    for fr, r0, r1 in slabs:
        conv = torch.from_numpy(oracle.conv3x3(xb[fr:fr + 1, :, 2 * r0:2 * (r1 - 1) + 3].double().cpu().numpy(), w.double().cpu().numpy(), stride=2))
        act = oracle.bias_act(conv, b.double().cpu(), act='lrelu', alpha=alpha, gain=gain)
        ref = act + (res0[fr:fr + 1, :, r0:r1].double().cpu() if with_res else 0.0)
        scale = act.abs().max().item()
        err = (y[fr:fr + 1, :, r0:r1].double().cpu() - ref).abs().max().item() / scale
        assert err < 1e-5, f'frame {fr} rows {r0}:{r1}: relative error {err:.2e} vs the float64 oracle composition'
        if with_res:
            assert (a[fr:fr + 1, :, r0:r1].double().cpu() - act).abs().max().item() / scale < 1e-5, 'activation output stored beside the sum'
    xi = torch.randint(-3, 4, xb.shape, generator=g, device=DEV).float()
    wi = torch.randint(-2, 3, w.shape, generator=g, device=DEV).float()
    bi = torch.randint(-40, 41, [cs], generator=g, device=DEV).float()
    ri = torch.randint(-100, 101, [n, cs, hs, hs], generator=g, device=DEV).float() if with_res else None
    yi, ai = fused_down_act._launch(xi, wi, bi, ri.clone() if with_res else None, with_res, 3, ALPHA, GAIN, -1.0)
    plain = conv2d_gradfix._native_conv(xi, wi, S2)
    act_i = _lrelu_formula(plain, None, bi, ALPHA, GAIN)
    want = act_i + ri if with_res else act_i
    assert torch.equal(yi, want), f'{int((yi != want).sum())} of {yi.numel()} elements of the whole fused output differ from epilogue(plain strided convolution) (+ residual)'
    if with_res:
        assert torch.equal(ai, act_i)
    fr, r0, r1 = slabs[1]
    convi = torch.from_numpy(oracle.conv3x3(xi[fr:fr + 1, :, 2 * r0:2 * (r1 - 1) + 3].double().cpu().numpy(), wi.double().cpu().numpy(), stride=2))
    refi = oracle.bias_act(convi, bi.double().cpu(), act='lrelu', alpha=ALPHA, gain=GAIN) + (ri[fr:fr + 1, :, r0:r1].double().cpu() if with_res else 0.0)
    assert torch.equal(yi[fr:fr + 1, :, r0:r1].double().cpu(), refi), 'integer slab vs the oracle composition is not exact'


# ---------------------------------------------------------------------------------------------------------------------------------------
# sgv_upfirdn2d_fused modes 1-3 at [96, 64, 257, 257] <-> [96, 64, 256, 256]

def _planes():
    return [(0, 0), (N // 2 + 1, 31), (N - 1, 63)]


def _dense(t):
    """CPU copy with canonical strides (oracle.bias_act compares strides; a [1, 1, H, W] slice of a batch keeps the batch's strides in its size-1 dims)."""
    t = t.detach().cpu()
    return t.reshape(-1).clone().reshape(t.shape)


def test_fused_fir_epilogue_modes_1_and_2_at_benchmark_shape():
    """The up-sampling SynthesisLayer's tail (upfirdn2d -> * dcoefs -> bias_act, conv2d_resample.py:138-139 + networks.py:70-74,141-143) as one
    kernel (mode 1) and its backward as one kernel (mode 2), b256.conv0: [96, 64, 257, 257] -> [96, 64, 256, 256]."""
    g = torch.Generator(device=DEV).manual_seed(257)
    c = 64
    x = torch.randn([N, c, 257, 257], generator=g, device=DEV).requires_grad_(True)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(DEV)
    sc = (torch.rand([N, c], generator=g, device=DEV) + 0.5).requires_grad_(True)
    b = (torch.randn([c], generator=g, device=DEV) * 0.5).requires_grad_(True)
    custom_ops.prof_enable(64)
    y = fused_fir_act.fir_bias_act(x, f, scale=sc, bias=b, padding=1, fir_gain=4, act='lrelu')
    custom_ops.prof_disable()
    prof = custom_ops.prof_collect()
    dispatch_assert(prof['upfirdn2d_lanes']['launches'] == 1 and prof['bias_act']['launches'] == 0 and prof['modulate']['launches'] == 0, 'forward must be ONE kernel')
    assert y.shape == (N, c, 256, 256)
    yc = fused_fir_act.fir_bias_act_composed(x, f, scale=sc, bias=b, padding=1, fir_gain=4, act='lrelu')
    assert torch.equal(y, yc), 'mode 1 differs from the three-pass composition of this library (fp32: same operations, same order)'
    for n, ch in _planes():
        u = oracle.upfirdn2d(_dense(x[n:n + 1, ch:ch + 1]), f.cpu(), padding=1, gain=4)
        ref = oracle.bias_act(_dense(u * sc[n, ch].detach().cpu()), b[ch:ch + 1].detach().cpu(), act='lrelu')
        assert torch.equal(y[n:n + 1, ch:ch + 1].detach().cpu(), ref), f'plane ({n}, {ch}): mode 1 is not bit-exact vs oracle.upfirdn2d -> * scale -> oracle.bias_act'

    dy = torch.randn(y.shape, generator=g, device=DEV)
    custom_ops.prof_enable(64)
    gx, gs, gb = torch.autograd.grad(y, [x, sc, b], dy)
    custom_ops.prof_disable()
    prof = custom_ops.prof_collect()
    dispatch_assert(prof['upfirdn2d_lanes']['launches'] == 1 and prof['bias_act']['launches'] == 0, 'backward must be ONE kernel')
    cx, cs_, cb_ = torch.autograd.grad(yc, [x, sc, b], dy)
    assert torch.equal(gx, cx), 'mode 2: input gradient differs from the composition'
    for got, want, name in ((gs, cs_, 'scale'), (gb, cb_, 'bias')):
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err < 2e-5, f'mode 2 {name} gradient (in-kernel plane sums): {err:.2e}'
    for n, ch in _planes():
        yp = _dense(y[n:n + 1, ch:ch + 1])
        gz = oracle.bias_act(_dense(dy[n:n + 1, ch:ch + 1]), None, act='lrelu', grad=1, xref=yp, yref=yp)     # lrelu: the sign of y selects the branch
        ref = oracle.upfirdn2d(_dense(gz * sc[n, ch].detach().cpu()), f.cpu(), padding=2, flip_filter=True, gain=4)
        assert torch.equal(gx[n:n + 1, ch:ch + 1].cpu(), ref), f'plane ({n}, {ch}): mode 2 is not bit-exact vs the oracle composition'


def test_fused_fir_backward_epilogue_mode_3_at_benchmark_shape():
    """Gradient of "bias_act, then the FIR in front of the strided convolution" (DiscriminatorBlock conv0 -> conv1 of b256) as one kernel:
    g [96, 64, 257, 257] -> FIR-transposed -> lrelu'(y0) -> dz [96, 64, 256, 256], with the bias-gradient plane sums."""
    g = torch.Generator(device=DEV).manual_seed(258)
    c = 64
    lib = custom_ops.get_native()
    gin = torch.randn([N, c, 257, 257], generator=g, device=DEV)
    y0 = torch.randn([N, c, 256, 256], generator=g, device=DEV)       # the saved forward output (only its sign / saturation matters)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(DEV)
    alpha, gain = 0.2, 2 ** 0.5
    dz = torch.empty_like(y0)
    sums = torch.zeros([N * c], dtype=torch.float32, device=DEV)
    bpads = (4 - 2 - 1, 256 - 257 + 2, 4 - 2 - 1, 256 - 257 + 2)
    e = custom_ops.FirEpilogue(3, None, None, y0.data_ptr(), sums.data_ptr(), None, 3, alpha, gain, -1.0)
    with custom_ops.device_guard(gin):
        custom_ops.check(lib.sgv_upfirdn2d_fused(fused_fir_act._ufd_params(gin, f, dz, bpads, True, 1.0), e, 0, custom_ops.raw_stream(gin)), lib)
    gy = upfirdn2d.upfirdn2d(gin, f, padding=list(bpads), flip_filter=True)
    want = torch.where(y0 > 0, gy, gy * alpha) * gain
    assert torch.equal(dz, want), 'mode 3 differs from FIR pass + activation gradient as separate passes'
    ps = want.sum(dim=(2, 3), dtype=torch.float64).reshape(-1)
    assert (sums.double() - ps).abs().max().item() / ps.abs().max().item() < 2e-5, 'in-kernel plane sums'
    for n, ch in _planes():
        u = _dense(oracle.upfirdn2d(_dense(gin[n:n + 1, ch:ch + 1]), f.cpu(), padding=list(bpads), flip_filter=True))
        yp = _dense(y0[n:n + 1, ch:ch + 1])
        ref = oracle.bias_act(u, None, act='lrelu', alpha=alpha, gain=gain, grad=1, xref=yp, yref=yp)
        assert torch.equal(dz[n:n + 1, ch:ch + 1].cpu(), ref), f'plane ({n}, {ch}): mode 3 is not bit-exact vs oracle.upfirdn2d -> oracle.bias_act(grad=1)'
