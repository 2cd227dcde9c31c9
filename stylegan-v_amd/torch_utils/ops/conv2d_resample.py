"""2-D convolution with optional FIR up/down-sampling.

Host-side mirror of the reference's ``src/torch_utils/ops/conv2d_resample.py`` (``conv2d_resample``
:59).  It only decides WHICH kernels run (``plan``: pure geometry, a list of steps): the convolutions go
through ``conv2d_gradfix`` -- this library's MFMA kernels for every 3x3 / 1x1 shape of the path, the vendor
library for what they do not serve --; every resampling step is an ``upfirdn2d`` launch
(csrc/upfirdn2d.hip).  The decomposition per case is the one of conv2d_resample.py:107-154:

  1x1 conv, down > 1 .......... FIR+decimate first, then convolve on the small image
  1x1 conv, up > 1 ............ convolve first, then zero-insert+FIR
  kxk conv, down > 1 .......... FIR at full resolution, then a stride-`down` convolution
  kxk conv, up > 1 ............ stride-`up` transposed convolution, then FIR (gain up^2)
  no resampling, symmetric non-negative padding ... plain convolution
  anything else ............... upfirdn2d -> convolution -> upfirdn2d
"""

import torch

from .. import misc
from . import conv2d_gradfix
from . import gemm as _gemm
from . import pointwise as _pw
from . import upfirdn2d as _ufd
from .upfirdn2d import _get_filter_size, _parse_padding


def _get_weight_shape(w):
    return [int(s) for s in w.shape]


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d / conv_transpose2d.  ``flip_weight=True`` is correlation (what ``F.conv2d`` does);
    ``False`` means true convolution, obtained by flipping the kernel spatially."""
    out_ch, in_ch_per_group, kh, kw = _get_weight_shape(w)
    if not flip_weight and (kh > 1 or kw > 1):
        from . import amax as _amax
        w = _amax.same_values(w.flip([2, 3]), w) if (w.is_cuda and w.dtype == torch.float32) else w.flip([2, 3])
    # 1x1 convolution with <= 4 channels on one side (fromRGB, un-modulated ToRGB): a memory stream, not a GEMM.
    if kh == 1 and kw == 1 and stride == 1 and padding in (0, [0, 0], (0, 0)) and not transpose and groups == 1 \
            and min(out_ch, in_ch_per_group) <= 4 and _pw.enabled and x.is_cuda and x.is_contiguous() and (x.shape[2] * x.shape[3]) % 4 == 0:
        return _pw.pointwise_conv(x, w.reshape(1, out_ch, in_ch_per_group))
    # Dense fp32 1x1 convolution on whole 128x128 tiles (the discriminator's skip branches): own MFMA GEMM on NCHW.
    if kh == 1 and kw == 1 and stride == 1 and padding in (0, [0, 0], (0, 0)) and not transpose and groups == 1 and _gemm.enabled \
            and w.dtype == torch.float32 and _gemm.is_full_tile_conv1x1(x, out_ch):
        return _gemm.conv1x1(x, w)
    # channels_last 1x1 convolutions with few channels are a plain matrix product (conv2d_resample.py:40-50).
    if kh == 1 and kw == 1 and stride == 1 and padding in (0, [0, 0], (0, 0)) and not transpose:
        if x.stride(1) == 1 and min(out_ch, in_ch_per_group) < 64:
            if out_ch <= 4 and groups == 1:
                n, _, h, wd = x.shape
                y = w.reshape(out_ch, in_ch_per_group).to(x.dtype) @ x.reshape(n, in_ch_per_group, -1)
                y = y.reshape(n, out_ch, h, wd)
            else:
                y = conv2d_gradfix.conv2d(x.contiguous(), w.contiguous(), groups=groups)
            return y.to(memory_format=torch.channels_last)
    op = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return op(x, w, stride=stride, padding=padding, groups=groups)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Geometry.  Everything below works per AXIS on a pair (lo, hi) of paddings; a "plan" is the list of kernel launches a call decomposes into.

def _footprint(pad, taps, up, down):
    """Padding of one axis with the resampling filter's own footprint folded in (what ``upsample2d`` / ``downsample2d`` add for a `taps`-tap filter)."""
    lo, hi = pad
    if up > 1:
        lo, hi = lo + (taps + up - 1) // 2, hi + (taps - up) // 2
    if down > 1:
        lo, hi = lo + (taps - down + 1) // 2, hi + (taps - down) // 2
    return lo, hi


def _axes(f, up, down, padding):
    """((x_lo, x_hi), (y_lo, y_hi)) for filter f."""
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)
    return _footprint((px0, px1), fw, up, down), _footprint((py0, py1), fh, up, down)


def _flat(ax, ay):
    return [ax[0], ax[1], ay[0], ay[1]]


def _transposed_axis(pad, k, up):
    """A stride-`up` transposed convolution with a k-tap kernel covers k - 1 (lo) and k - up (hi) of an axis' padding by itself; what is then negative is
    cropped by the convolution (its `padding` argument: the part both ends share), the remainder stays with the FIR pass behind it.
    -> (convolution padding, FIR (lo, hi))."""
    lo, hi = pad[0] - (k - 1), pad[1] - (k - up)
    crop = max(min(-lo, -hi), 0)
    return crop, (lo + crop, hi + crop)


def _transposed_weight(w, groups):
    """[Cout, Cin/g, kh, kw] -> the [Cin, Cout/g, kh, kw] layout conv_transpose2d wants."""
    out_ch, in_ch_per_group, kh, kw = _get_weight_shape(w)
    if groups == 1:
        if w.is_cuda and w.dtype == torch.float32:
            from . import amax as _amax
            return _amax.same_values(w.transpose(0, 1), w)      # (the view's magnitude bound is the parameter's: taken once per optimiser step, not per call)
        return w.transpose(0, 1)
    wt = w.reshape(groups, out_ch // groups, in_ch_per_group, kh, kw).transpose(1, 2)
    return wt.reshape(groups * in_ch_per_group, out_ch // groups, kh, kw)


def downsampling_pads(f, down, padding=0):
    """(px0, px1, py0, py1) of the FIR pass in front of a stride-`down` convolution."""
    return tuple(_flat(*_axes(f, 1, down, padding)))


def downsampling_filter_pass(x, f, down, padding=0, kernel_hw=(3, 3), flip_filter=False):
    """The FIR pass that ``conv2d_resample`` puts in front of its strided convolution (`down > 1 and up == 1`, non-pointwise kernel): returns the
    filtered tensor the stride-`down` convolution then reads, so that a caller can run a fused convolution tail on it (ops/fused_down_act.py)."""
    return _ufd.upfirdn2d(x=x, f=f, padding=list(downsampling_pads(f, down, padding)), flip_filter=flip_filter)


def downsampling_conv1x1(x, w, f, down, padding=0, residual=None, flip_filter=False, prefiltered=False):
    """``conv2d_resample`` for a 1x1 kernel with ``down > 1`` (FIR + decimate, then convolve on the small image: the skip branch of the
    residual discriminator block) with the other branch's result added in the convolution's store where the MFMA GEMM serves the shape;
    the fallback adds out of place (the residual is another layer's activation output, which its backward pass still needs)."""
    out_ch, in_ch, kh, kw = _get_weight_shape(w)
    assert kh == 1 and kw == 1 and down > 1
    if not prefiltered:   # (prefiltered: the caller already ran this FIR + decimate pass, e.g. fused_fir_act.fir_down_with_input_alias)
        x = _ufd.upfirdn2d(x=x, f=f, down=down, padding=list(downsampling_pads(f, down, padding)), flip_filter=flip_filter)
    if residual is not None and _gemm.enabled and w.dtype == torch.float32 and x.is_cuda and _gemm.is_full_tile_conv1x1(x, out_ch) \
            and residual.dtype == torch.float32 and tuple(residual.shape) == (x.shape[0], out_ch, x.shape[2], x.shape[3]):
        return _gemm.conv1x1(x, w, residual=residual)
    y = _conv2d_wrapper(x=x, w=conv2d_gradfix.cast_weight(w, x))
    return y + residual if residual is not None else y


def upsampling_conv_parts(x, w, f, up, padding=0, groups=1, flip_weight=True):
    """First half of the `up > 1` decomposition: the stride-`up` transposed convolution, WITHOUT the FIR that follows it.  Returns (y, fir_padding):
    ``upfirdn2d(y, f, padding=fir_padding, gain=up**2)`` completes the op.  Used by the synthesis layer to fuse that FIR with the demodulation / bias /
    activation epilogue."""
    assert up > 1
    _, _, kh, kw = _get_weight_shape(w)
    ax, ay = _axes(f, up, 1, padding)
    cx, fx = _transposed_axis(ax, kw, up)
    cy, fy = _transposed_axis(ay, kh, up)
    y = _conv2d_wrapper(x=x, w=_transposed_weight(w, groups), stride=up, padding=[cy, cx], groups=groups, transpose=True, flip_weight=(not flip_weight))
    return y, _flat(fx, fy)


def plan(kernel_hw, f, up=1, down=1, padding=0):
    """The launches ``conv2d_resample`` decomposes a call into, as a list of (kind, arguments) steps -- the case table of the module docstring, one row each:

      kind 'fir'  : upfirdn2d(x, f or None, up, down, padding, gain = up ** 2)
      kind 'conv' : convolution (stride, padding [y, x]); 'convT': transposed convolution on the transposed weight

    Pure geometry: no tensors, so the selection is testable on its own (tests/test_ops_cpu.py)."""
    kh, kw = kernel_hw
    ax, ay = _axes(f, up, down, padding)
    pointwise = kh == 1 and kw == 1
    fir = lambda pads, u=1, d=1, with_filter=True: ('fir', dict(up=u, down=d, padding=pads, gain=u ** 2, with_filter=with_filter))      # noqa: E731
    conv = lambda stride=1, pad=(0, 0): ('conv', dict(stride=stride, padding=list(pad)))                                              # noqa: E731
    if up == 1 and down > 1:
        # the decimation commutes with a 1x1 kernel (filter on the way down, convolve the small image); a k x k kernel strides over the filtered full-size image
        return [fir(_flat(ax, ay), d=down), conv()] if pointwise else [fir(_flat(ax, ay)), conv(stride=down)]
    if up > 1 and down == 1 and pointwise:
        return [conv(), fir(_flat(ax, ay), u=up)]                 # convolve the small image, then zero-insert + filter
    if up > 1:
        cx, fx = _transposed_axis(ax, kw, up)
        cy, fy = _transposed_axis(ay, kh, up)
        steps = [('convT', dict(stride=up, padding=[cy, cx])), ('fir', dict(up=1, down=1, padding=_flat(fx, fy), gain=up ** 2, with_filter=True))]
        return steps + ([fir([0, 0, 0, 0], d=down)] if down > 1 else [])
    if ax[0] == ax[1] and ay[0] == ay[1] and ax[0] >= 0 and ay[0] >= 0:
        return [conv(pad=(ay[0], ax[0]))]                         # no resampling, padding the convolution can apply itself
    return [fir(_flat(ax, ay), with_filter=False), conv()]       # ragged / negative padding: pad (or crop) with a filterless pass first


@misc.profiled_function
def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """x ``[N, Cin, H, W]``, w ``[Cout, Cin/groups, kh, kw]``, f from ``upfirdn2d.setup_filter``.
    ``padding`` is relative to the upsampled image and applied once, up front.
    ``flip_weight=True`` = correlation, ``flip_filter=False`` = convolution (conv2d_resample.py:59-83)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype in (x.dtype, torch.float32)   # fp32 master weights next to 16-bit activations: conv2d_gradfix.cast_weight
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1
    assert isinstance(down, int) and down >= 1
    assert isinstance(groups, int) and groups >= 1
    _, _, kh, kw = _get_weight_shape(w)
    for kind, a in plan((kh, kw), f, up=up, down=down, padding=padding):
        if kind == 'fir':
            x = _ufd.upfirdn2d(x=x, f=(f if a['with_filter'] else None), up=a['up'], down=a['down'], padding=a['padding'], gain=a['gain'], flip_filter=flip_filter)
        elif kind == 'conv':
            x = _conv2d_wrapper(x=x, w=w, stride=a['stride'], padding=a['padding'], groups=groups, flip_weight=flip_weight)
        else:
            x = _conv2d_wrapper(x=x, w=_transposed_weight(w, groups), stride=a['stride'], padding=a['padding'], groups=groups, transpose=True, flip_weight=(not flip_weight))
    return x
