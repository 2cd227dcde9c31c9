"""The 3x3 convolution family at the shapes the headline benchmark runs (BASELINE config 3: 32 videos x 3 frames = 96
frames, FFS-256 channel ladder) against oracle/oracle.py, through the C ABI.

Every launch here has thousands of tiles on 256 persistent workgroups, so the cross-tile prefetch, the XCD-aware tile order
and the multi-unit atomics flush of the weight-gradient kernels are what is being tested (the small cases of
test_conv3x3_gpu.py / test_conv_wrw_gpu.py run a handful of tiles).  The float64 oracle cannot afford a whole 464-GFLOP
layer, so each case combines

  * slab checks: output rows of a few frames (first / middle / last frame, top / interior / bottom rows, all channels)
    against the oracle on random data -- relative error < 1e-5 (north_star: 1e-3; bf16x3 sits at ~4e-6);
  * the same slabs on small-integer data: exact (the hi/lo split, the products and the fp32 sums are exact there), which pins
    tap / channel / pixel indexing;
  * a full-tensor checksum on integer data: the per-(frame, channel) plane sums of the WHOLE output must equal, exactly,
    what linearity predicts from window sums of the input (a tile that is skipped, written twice or mis-placed between
    planes changes a plane sum);
  * weight gradients: a sampled (out, in) channel block on random data, and on integer data the exact row / column checksums
    sum_o dw[o, i] and sum_i dw[o, i] over ALL channels and ALL frames (1-channel problems by linearity, evaluated in
    float64 on the device by a nine-pass restatement that is itself pinned to the oracle on two frames).
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
N = 96   # frames per GPU of the benchmark step


def _cfg(stride, transposed):
    return (transposed, (stride, stride), (1, 1) if stride == 1 else (0, 0), (0, 0), (1, 1), 1)


def _run(x, w, stride, transposed):
    cfg = _cfg(stride, transposed)
    dispatch_assert(conv2d_gradfix._native_conv_ok(x, w, cfg), 'the benchmark shape is not served by the hand-written kernel')
    custom_ops.prof_enable(16)
    y = conv2d_gradfix._native_conv(x, w, cfg)
    custom_ops.prof_disable()
    dispatch_assert(custom_ops.prof_collect()['conv3x3']['launches'] == 1)
    return y


def _slab_ref(x, w, stride, transposed, n, r0, r1):
    """Oracle rows [r0, r1) of frame n of the output (all channels), from exactly the input rows they depend on."""
    h = x.shape[2]
    wn = w.double().cpu().numpy()
    if stride == 1:            # forward and transposed stride-1 / pad-1: output row Y reads input rows Y-1 .. Y+1
        lo, hi = max(r0 - 1, 0), min(r1 + 1, h)
        ref = oracle.conv3x3(x[n:n + 1, :, lo:hi].cpu().numpy(), wn, stride=1, transposed=transposed)
        return ref[:, :, r0 - lo:r0 - lo + (r1 - r0)]
    if not transposed:         # strided: output row Y reads input rows 2Y .. 2Y+2
        return oracle.conv3x3(x[n:n + 1, :, 2 * r0:2 * (r1 - 1) + 3].cpu().numpy(), wn, stride=2)
    # transposed stride 2: output row R receives input rows Y with 2Y <= R <= 2Y + 2
    lo, hi = max((r0 - 1) // 2, 0), min((r1 - 1) // 2, h - 1) + 1
    ref = oracle.conv3x3(x[n:n + 1, :, lo:hi].cpu().numpy(), wn, stride=2, transposed=True)
    return ref[:, :, r0 - 2 * lo:r0 - 2 * lo + (r1 - r0)]


def _plane_sums_ref(x, w, stride, transposed, out_hw):
    """Exact per-(frame, out channel) sums of the output by linearity (integer data, float64 on the device)."""
    n, k, h, wd = x.shape
    ho, wo = out_hw
    pad = 1 if stride == 1 else 0
    a = torch.zeros([n, k, 3, 3], dtype=torch.float64, device=x.device)
    for ky in range(3):
        for kx in range(3):
            if not transposed:    # tap (ky,kx) reads x[sY+ky-p, sX+kx-p] for every output (Y, X) that stays inside the image
                ys = [s for s in range(ky - pad, ky - pad + stride * (ho - 1) + 1, stride) if 0 <= s < h]
                xs = [s for s in range(kx - pad, kx - pad + stride * (wo - 1) + 1, stride) if 0 <= s < wd]
            else:                 # tap scatters x[Y, X] to (sY+ky-p, sX+kx-p): the inputs whose target stays inside the output
                ys = [s for s in range(h) if 0 <= stride * s + ky - pad < ho]
                xs = [s for s in range(wd) if 0 <= stride * s + kx - pad < wo]
            sub = x[:, :, ys[0]:ys[-1] + 1:(ys[1] - ys[0] if len(ys) > 1 else 1), xs[0]:xs[-1] + 1:(xs[1] - xs[0] if len(xs) > 1 else 1)]
            a[:, :, ky, kx] = sub.sum(dim=(2, 3), dtype=torch.float64)
    wd64 = w.double()
    return torch.einsum('nkab,mkab->nm', a, wd64) if not transposed else torch.einsum('nkab,kmab->nm', a, wd64)


def _check_conv(ci, co, h, wd, stride, transposed, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn([N, ci, h, wd], generator=g, device=DEV) + 0.25
    w = torch.randn([ci, co, 3, 3] if transposed else [co, ci, 3, 3], generator=g, device=DEV) / (3 * ci ** 0.5)
    xi = torch.randint(-3, 4, x.shape, generator=g, device=DEV).float()
    wi = torch.randint(-2, 3, w.shape, generator=g, device=DEV).float()
    y, yi = _run(x, w, stride, transposed), _run(xi, wi, stride, transposed)
    ho, wo = y.shape[2:]
    rows = 6 if h >= 64 else min(ho, 8)
    slabs = [(0, 0, rows), (N // 2 + 1, (ho - rows) // 2 + 1, (ho - rows) // 2 + 1 + rows), (N - 1, ho - rows, ho)]   # top of the first, interior (odd start) of a middle, bottom of the last frame
    worst = 0.0
    for n, r0, r1 in slabs:
        ref = _slab_ref(x, w, stride, transposed, n, r0, r1)
        got = y[n:n + 1, :, r0:r1].double().cpu().numpy()
        assert got.shape == ref.shape
        err = np.abs(got - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        assert err < 1e-5, f'frame {n} rows {r0}:{r1}: relative error {err:.2e} vs the float64 oracle'
        refi = _slab_ref(xi, wi, stride, transposed, n, r0, r1)
        assert np.array_equal(yi[n:n + 1, :, r0:r1].double().cpu().numpy(), refi), f'frame {n} rows {r0}:{r1}: integer data is not exact'
    sums = yi.sum(dim=(2, 3), dtype=torch.float64)
    want = _plane_sums_ref(xi, wi, stride, transposed, (ho, wo))
    assert torch.equal(sums, want), f'{int((sums != want).sum())} of {sums.numel()} output planes have a wrong checksum'
    print(f'[{ci}->{co} {h}x{wd} s{stride}{"T" if transposed else ""}] worst slab error {worst:.2e}')


@pytest.mark.parametrize('c,r', [(64, 256), (128, 128), (256, 64), (512, 32), (512, 16), (512, 8)])
def test_conv3x3_s1_forward_at_benchmark_shapes(c, r):
    """SynthesisLayer conv1 / DiscriminatorBlock conv0 of every block (networks.py:141, 470) incl. the 16^2 / 8^2 kernel."""
    _check_conv(c, c, r, r, 1, False, seed=c + r)


@pytest.mark.parametrize('c,r', [(64, 256), (512, 32), (512, 16)])
def test_conv3x3_s1_data_gradient_at_benchmark_shapes(c, r):
    """Their data gradients: the transposed stride-1 form with re-indexed weights (conv2d_gradfix.py:100-118)."""
    _check_conv(c, c, r, r, 1, True, seed=2 * c + r)


@pytest.mark.parametrize('cb,cs,hs', [(64, 128, 128), (128, 256, 64), (256, 512, 32)])
def test_conv3x3_strided_at_benchmark_shapes(cb, cs, hs):
    """DiscriminatorBlock conv1 after its FIR (conv2d_resample.py:119-122): [96, cb, 2hs+1, 2hs+1] -> [96, cs, hs, hs]."""
    _check_conv(cb, cs, 2 * hs + 1, 2 * hs + 1, 2, False, seed=cb + hs)


@pytest.mark.parametrize('cs,cb,hs', [(512, 256, 32), (256, 128, 64), (128, 64, 128)])
def test_conv3x3_transposed_at_benchmark_shapes(cs, cb, hs):
    """SynthesisLayer conv0 before its FIR (conv2d_resample.py:125-137): [96, cs, hs, hs] -> [96, cb, 2hs+1, 2hs+1]."""
    _check_conv(cs, cb, hs, hs, 2, True, seed=cs + hs + 1)


def _wrw_checksum(small, big, stride):
    """dw[s, b, ky, kx] = sum_{n,Y,X} small[n,s,Y,X] * big[n,b,stride*Y+ky-p,stride*X+kx-p] where one of the two tensors has
    a single channel: nine masked multiply-reduce passes in float64 on the device (exact on integer data)."""
    pad = 1 if stride == 1 else 0
    hs, ws = small.shape[2:]
    bp = torch.nn.functional.pad(big, (pad, pad, pad, pad))
    out = torch.zeros([small.shape[1], big.shape[1], 3, 3], dtype=torch.float64, device=small.device)
    for ky in range(3):
        for kx in range(3):
            win = bp[:, :, ky:ky + stride * (hs - 1) + 1:stride, kx:kx + stride * (ws - 1) + 1:stride]
            prod = (win * small).sum(dim=(0, 2, 3), dtype=torch.float64)     # one operand has 1 channel: broadcasts to the other's channels
            out[:, :, ky, kx] = prod.reshape(out.shape[0], out.shape[1])
    return out


def _check_wrw(c_small, c_big, hs, stride, transposed, seed):
    """Weight gradient of a layer mapping (stride 1) c_big -> c_small channels at hs x hs, or of the stride-2 pair."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    hb = hs if stride == 1 else 2 * hs + 1
    if transposed:      # layer input is the small tensor, dy the big one; w is [c_small, c_big, 3, 3]
        x_shape, dy_shape = [N, c_small, hs, hs], [N, c_big, hb, hb]
    else:               # layer input is the big tensor, dy the small one; w is [c_small, c_big, 3, 3]
        x_shape, dy_shape = [N, c_big, hb, hb], [N, c_small, hs, hs]
    w_shape = (c_small, c_big, 3, 3)
    cfg = _cfg(stride, transposed)

    def run(dy, x):
        dispatch_assert(conv2d_gradfix._native_wrw_ok(dy, x, cfg, w_shape), 'the benchmark shape is not served by the hand-written kernel')
        custom_ops.prof_enable(16)
        dw = conv2d_gradfix._native_wrw(dy, x, cfg, w_shape)
        custom_ops.prof_disable()
        dispatch_assert(custom_ops.prof_collect()['conv_wrw']['launches'] == 1)
        return dw

    # (1) random data, sampled channel block
    x = torch.randn(x_shape, generator=g, device=DEV) * 1.5 + 0.25
    dy = torch.randn(dy_shape, generator=g, device=DEV)
    dw = run(dy, x)
    so = torch.tensor([0, 1, c_small // 2 + 3, c_small - 1])      # channels of the small-side tensor
    sb = torch.tensor([0, 31, c_big // 2 + 5, c_big - 1])         # channels of the big-side tensor
    xs, dys = (x[:, so], dy[:, sb]) if transposed else (x[:, sb], dy[:, so])
    ref = oracle.conv3x3_weight_grad(dys.cpu().numpy(), xs.cpu().numpy(), stride=stride, transposed=transposed)
    got = dw[so][:, sb].double().cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-5, f'sampled weight-gradient block: relative error {err:.2e} vs the float64 oracle'

    # (2) integer data in {-1, 0, 1} (|dw| <= 9 * N * H * W < 2^24: every partial sum and every atomic add is exact)
    xi = torch.randint(-1, 2, x_shape, generator=g, device=DEV).float()
    dyi = torch.randint(-1, 2, dy_shape, generator=g, device=DEV).float()
    dwi = run(dyi, xi).double()
    assert torch.equal(dwi, dwi.round())
    small_i, big_i = (xi, dyi) if transposed else (dyi, xi)
    col = _wrw_checksum(small_i.sum(1, keepdim=True), big_i, stride)      # [1, c_big, 3, 3]   = sum over the small-side channels
    row = _wrw_checksum(small_i, big_i.sum(1, keepdim=True), stride)      # [c_small, 1, 3, 3] = sum over the big-side channels
    assert torch.equal(dwi.sum(0, keepdim=True), col), 'column checksum of the integer weight gradient is not exact'
    assert torch.equal(dwi.sum(1, keepdim=True), row), 'row checksum of the integer weight gradient is not exact'
    # and the checksum helper itself against the oracle on the first two frames (it is a restatement, so it gets pinned too)
    o_args = (big_i[:2].sum(1, keepdim=True).cpu().numpy(), small_i[:2].cpu().numpy()) if transposed else (small_i[:2].cpu().numpy(), big_i[:2].sum(1, keepdim=True).cpu().numpy())
    assert np.array_equal(_wrw_checksum(small_i[:2], big_i[:2].sum(1, keepdim=True), stride).cpu().numpy(),
                          oracle.conv3x3_weight_grad(*o_args, stride=stride, transposed=transposed))
    print(f'[wrw {c_big}->{c_small} {hs} s{stride}{"T" if transposed else ""}] sampled block error {err:.2e}')


@pytest.mark.parametrize('c,r', [(64, 256), (128, 128), (256, 64), (512, 32), (512, 16), (512, 8)])
def test_conv3x3_weight_gradient_at_benchmark_shapes(c, r):
    """Conv2dGradWeight of the stride-1 layers (conv2d_gradfix.py:140-170)."""
    _check_wrw(c, c, r, 1, False, seed=3 * c + r)


@pytest.mark.parametrize('cs,cb,hs,transposed', [(128, 64, 128, False), (512, 256, 32, False), (512, 256, 32, True), (128, 64, 128, True),
                                                  (512, 512, 16, False), (512, 512, 16, True), (512, 512, 8, False), (512, 512, 8, True)])
def test_conv3x3_stride2_weight_gradient_at_benchmark_shapes(cs, cb, hs, transposed):
    """... of the strided (D) and transposed (G) stride-2 layers; the weight is [c_small, c_big, 3, 3] in both."""
    _check_wrw(cs, cb, hs, 2, transposed, seed=5 * cs + hs + transposed)
