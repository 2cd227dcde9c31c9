"""upfirdn2d: pad -> zero-insert (up) -> 2-D FIR -> decimate (down), for batches of images.

Host-side mirror of the reference op ``src/torch_utils/ops/upfirdn2d.py`` (same public names,
argument meaning, defaults and error behaviour): ``setup_filter`` (:72), ``upfirdn2d`` (:120),
``filter2d`` (:272), ``upsample2d`` (:308), ``downsample2d`` (:347), ``_parse_padding`` (:46),
``_get_filter_size`` (:57).  GPU tensors run the hand-written gfx950 kernels of
``csrc/upfirdn2d.hip`` through the C ABI ``sgv_upfirdn2d`` (include/sgv_ops.h), which replaces the
reference's pybind entry point ``_plugin.upfirdn2d`` (upfirdn2d.cpp:16).

Dispatch: ``impl='cuda'`` on a GPU tensor -> native kernel, and a missing/broken native library is an
error (no silent fallback, unlike upfirdn2d.py:33-34).  CPU tensors or ``impl='ref'`` -> the plain
PyTorch formulation below (the reference's own CPU behaviour, upfirdn2d.py:162-164).
"""

import numpy as np
import torch
import torch.nn.functional as F

from .. import custom_ops
from .. import misc

_DTYPE_CODES = {torch.float32: custom_ops.SGV_F32, torch.float16: custom_ops.SGV_F16,
                torch.bfloat16: custom_ops.SGV_BF16, torch.float64: custom_ops.SGV_F64}

# ----------------------------------------------------------------------------------------------
# Argument helpers (same contracts as upfirdn2d.py:37-68).


def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = (scaling, scaling)
    assert isinstance(scaling, (list, tuple)) and len(scaling) == 2
    sx, sy = scaling
    assert isinstance(sx, int) and isinstance(sy, int)
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = (padding, padding)
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        return px, px, py, py
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _get_filter_size(f):
    """(width, height) of a filter tensor; (1, 1) for ``None``."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    assert fw >= 1 and fh >= 1
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Prepare a FIR filter for ``upfirdn2d`` (contract of upfirdn2d.py:72-116).

    ``f`` may be ``None`` (identity), a scalar, a 1-D tap list or a 2-D kernel.  1-D inputs with
    fewer than 8 taps become their 2-D outer product unless ``separable`` says otherwise.  Returns a
    float32 tensor: ``[taps]`` (separable) or ``[fh, fw]``.
    """
    taps = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert taps.ndim <= 2 and taps.numel() > 0
    if taps.ndim == 0:
        taps = taps.reshape(1)
    if separable is None:
        separable = taps.ndim == 1 and taps.numel() >= 8
    if taps.ndim == 1 and not separable:
        taps = torch.outer(taps, taps)
    assert taps.ndim == (1 if separable else 2)
    taps = taps.clone()
    if normalize:
        taps = taps / taps.sum()
    if flip_filter:
        taps = taps.flip(tuple(range(taps.ndim)))
    taps = taps * (gain ** (taps.ndim / 2))
    return taps.to(device=device)


def output_size(in_size, up, down, pad0, pad1, taps):
    """Output extent along one axis: C integer division, as upfirdn2d.cpp:32-33."""
    num = in_size * up + pad0 + pad1 - taps + down
    return int(num / down) if num < 0 else num // down  # truncate toward zero like C


# ----------------------------------------------------------------------------------------------
# Plain-PyTorch path (CPU tensors / impl='ref').


def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Slow PyTorch formulation (what upfirdn2d.py:169-208 computes): dense zero-stuffed signal,
    grouped ``conv2d`` with the (flipped) taps, then strided subsampling.  Differentiable to any order."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    assert f.dtype == torch.float32 and not f.requires_grad
    n, c, h, w = x.shape
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)

    if upx > 1 or upy > 1:  # zero insertion: sample (i, j) lands on (i*upy, j*upx)
        stuffed = x.new_zeros([n, c, h * upy, w * upx])
        stuffed[:, :, ::upy, ::upx] = x
        x = stuffed
    x = F.pad(x, [px0, px1, py0, py1])  # negative entries crop

    taps = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:  # conv2d correlates, so true convolution needs the flipped kernel
        taps = taps.flip(tuple(range(taps.ndim)))
    if taps.ndim == 2:
        x = F.conv2d(x, taps.expand(c, 1, *taps.shape), groups=c)
    else:  # separable: horizontal then vertical pass
        x = F.conv2d(x, taps.reshape(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
        x = F.conv2d(x, taps.reshape(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return x[:, :, ::downy, ::downx]


# ----------------------------------------------------------------------------------------------
# Native path.


def _native_call(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain, f_src=None):
    """One ``sgv_upfirdn2d`` launch on x's current stream.  Allocates and returns y.  ``f_src``: the caller's filter tensor when ``f2d`` is a fresh view of
    it (a separable pass): what the sum of its taps is remembered on (ops/amax.py)."""
    lib = custom_ops.get_native()
    if x.dtype not in _DTYPE_CODES:
        raise RuntimeError(f'upfirdn2d: unsupported dtype {x.dtype}')
    if not (f2d.is_cuda and f2d.device == x.device):
        raise RuntimeError('f must reside on the same device as x')
    if f2d.dtype != torch.float32:
        raise RuntimeError('f must be float32')
    if x.ndim != 4:
        raise RuntimeError('x must be rank 4')
    if f2d.ndim != 2:
        raise RuntimeError('f must be rank 2')
    n, c, h, w = x.shape
    fh, fw = f2d.shape
    ow = output_size(w, upx, downx, px0, px1, fw)
    oh = output_size(h, upy, downy, py0, py1, fh)
    if ow < 1 or oh < 1:
        raise RuntimeError('output must be at least 1x1')
    channels_last = x.stride(1) == 1 and c > 1 and x.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    if y.numel() == 0 or x.numel() == 0:
        return y
    xs, fs, ys = x.stride(), f2d.stride(), y.stride()
    p = custom_ops.Upfirdn2dParams(x.data_ptr(), f2d.data_ptr(), y.data_ptr(), upx, upy, downx, downy, px0, px1, py0, py1,
                                   int(bool(flip)), float(gain), w, h, c, n, xs[3], xs[2], xs[1], xs[0], fw, fh, fs[1], fs[0],
                                   ow, oh, ys[3], ys[2], ys[1], ys[0])
    with custom_ops.device_guard(x):
        from . import amax as _amax      # (a FIR output usually feeds a convolution: the LDS-tile kernel leaves its magnitude bound behind)
        f_rec = f2d if f_src is None else f_src
        if _amax.can_inherit_through_fir(x, f_rec):
            # |FIR(x)| <= gain * sum|taps| * bound(x): the output's bound follows from the input's without touching the tensor -- no side output armed, so no
            # atomics in the kernel and no fold launch behind it
            custom_ops.check(lib.sgv_upfirdn2d(p, _DTYPE_CODES[x.dtype], custom_ops.raw_stream(x)), lib)
        else:
            custom_ops.check(_amax.launch_tracking(y, lambda: lib.sgv_upfirdn2d(p, _DTYPE_CODES[x.dtype], custom_ops.raw_stream(x))), lib)
        _amax.inherit_through_fir(y, x, f_rec, gain)      # (no-op when the kernel left its own bound)
    return y


class _Upfirdn2dFn(torch.autograd.Function):
    """Autograd node for the native kernel.  cfg = (upx, upy, downx, downy, px0, px1, py0, py1, flip, gain).

    The gradient of upfirdn2d w.r.t. x is another upfirdn2d with up/down swapped, the filter flip
    inverted and the padding of upfirdn2d.py:251-261 -- so backward re-enters this same node and
    gradients of any order come for free.
    """

    @staticmethod
    def forward(ctx, x, f, cfg):
        upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = cfg
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            from . import amax as _amax
            _amax.set_tap_sum(f, 1.0)      # (known without reading the temporary back)
        assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
        if f.ndim == 2:
            y = _native_call(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain)
        else:  # separable taps: horizontal pass then vertical pass, gain split evenly (upfirdn2d.py:239-240)
            g = float(np.sqrt(gain))
            y = _native_call(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip, g, f_src=f)      # (sum |taps| of a view = that of f: remembered on f)
            y = _native_call(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip, g, f_src=f)
        ctx.cfg = cfg
        ctx.in_hw = (x.shape[2], x.shape[3])
        ctx.fsize = _get_filter_size(f)
        ctx.save_for_backward(f)
        return y

    @staticmethod
    def backward(ctx, dy):
        (f,) = ctx.saved_tensors
        upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = ctx.cfg
        ih, iw = ctx.in_hw
        oh, ow = dy.shape[2], dy.shape[3]
        fw, fh = ctx.fsize
        dx = None
        if ctx.needs_input_grad[0]:
            bcfg = (downx, downy, upx, upy,
                    fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1,
                    fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1,
                    not flip, gain)
            dx = _Upfirdn2dFn.apply(dy, f, bcfg)
        assert not ctx.needs_input_grad[1], 'upfirdn2d: the filter is not differentiable'
        return dx, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Pad, upsample, filter and downsample a batch of 2-D images (contract of upfirdn2d.py:120-164).

    x: ``[N, C, H, W]`` float16/bfloat16/float32/float64.  f: float32 ``[fh, fw]``, ``[taps]``
    (separable) or ``None``.  ``padding`` is relative to the upsampled image; negative = crop.
    ``flip_filter=False`` is true convolution.  Supports gradients of any order w.r.t. x.
    """
    assert isinstance(x, torch.Tensor)
    assert impl in ('ref', 'cuda')
    if impl == 'cuda' and x.device.type == 'cuda':
        upx, upy = _parse_scaling(up)
        downx, downy = _parse_scaling(down)
        cfg = (upx, upy, downx, downy) + _parse_padding(padding) + (bool(flip_filter), gain)
        return _Upfirdn2dFn.apply(x, f, cfg)
    return _upfirdn2d_ref(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)


# ----------------------------------------------------------------------------------------------
# Convenience wrappers: only padding arithmetic differs (upfirdn2d.py:296-304, 333-343, 372-382).


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """FIR-filter keeping the spatial size (plus user padding)."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    pads = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=pads, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Upsample by an integer factor; output size is ``up`` times the input (plus user padding)."""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    pads = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=pads, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Downsample by an integer factor; output size is the input divided by ``down`` (plus user padding)."""
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    pads = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=pads, flip_filter=flip_filter, gain=gain, impl=impl)
