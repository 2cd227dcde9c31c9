"""2-D convolution with optional FIR up/down-sampling.

Host-side mirror of the reference's ``src/torch_utils/ops/conv2d_resample.py`` (``conv2d_resample``
:59).  It only decides WHICH kernels run: the convolutions go to MIOpen through
``conv2d_gradfix``; every resampling step is an ``upfirdn2d`` launch (csrc/upfirdn2d.hip).  The
decomposition per case is the one of conv2d_resample.py:107-154:

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
        w = w.flip([2, 3])
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


def downsampling_conv1x1(x, w, f, down, padding=0, residual=None, flip_filter=False, prefiltered=False):
    """``conv2d_resample`` for a 1x1 kernel with ``down > 1`` (FIR + decimate, then convolve on the small image: the skip branch of the
    residual discriminator block) with the other branch's result added in the convolution's store where the MFMA GEMM serves the shape;
    the fallback adds out of place (the residual is another layer's activation output, which its backward pass still needs)."""
    out_ch, in_ch, kh, kw = _get_weight_shape(w)
    assert kh == 1 and kw == 1 and down > 1
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)
    pads = [px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2, py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2]
    if not prefiltered:   # (prefiltered: the caller already ran this FIR + decimate pass, e.g. fused_fir_act.fir_down_with_input_alias)
        x = _ufd.upfirdn2d(x=x, f=f, down=down, padding=pads, flip_filter=flip_filter)
    if residual is not None and _gemm.enabled and w.dtype == torch.float32 and x.is_cuda and _gemm.is_full_tile_conv1x1(x, out_ch) \
            and residual.dtype == torch.float32 and tuple(residual.shape) == (x.shape[0], out_ch, x.shape[2], x.shape[3]):
        return _gemm.conv1x1(x, w, residual=residual)
    y = _conv2d_wrapper(x=x, w=conv2d_gradfix.cast_weight(w, x))
    return y + residual if residual is not None else y


def downsampling_pads(f, down, padding=0):
    """(px0, px1, py0, py1) of the FIR pass in front of a stride-`down` convolution (the arithmetic of ``conv2d_resample`` below)."""
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)
    return (px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2, py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2)


def downsampling_filter_pass(x, f, down, padding=0, kernel_hw=(3, 3), flip_filter=False):
    """The FIR pass that ``conv2d_resample`` puts in front of its strided convolution (`down > 1 and up == 1`, non-pointwise kernel): returns the
    filtered tensor the stride-`down` convolution then reads, so that a caller can run a fused convolution tail on it (ops/fused_down_act.py)."""
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)
    px0 += (fw - down + 1) // 2
    px1 += (fw - down) // 2
    py0 += (fh - down + 1) // 2
    py1 += (fh - down) // 2
    return _ufd.upfirdn2d(x=x, f=f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)


def upsampling_conv_parts(x, w, f, up, padding=0, groups=1, flip_weight=True):
    """First half of the `up > 1` branch of conv2d_resample: the stride-`up` transposed convolution, WITHOUT the FIR
    that follows it.  Returns (y, fir_padding): ``upfirdn2d(y, f, padding=fir_padding, gain=up**2)`` completes the op.
    Used by the synthesis layer to fuse that FIR with the demodulation / bias / activation epilogue."""
    assert up > 1
    out_ch, in_ch_per_group, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)
    px0 += (fw + up - 1) // 2
    px1 += (fw - up) // 2
    py0 += (fh + up - 1) // 2
    py1 += (fh - up) // 2
    if groups == 1:
        wt = w.transpose(0, 1)
    else:
        wt = w.reshape(groups, out_ch // groups, in_ch_per_group, kh, kw).transpose(1, 2)
        wt = wt.reshape(groups * in_ch_per_group, out_ch // groups, kh, kw)
    px0 -= kw - 1
    px1 -= kw - up
    py0 -= kh - 1
    py1 -= kh - up
    pxt = max(min(-px0, -px1), 0)
    pyt = max(min(-py0, -py1), 0)
    y = _conv2d_wrapper(x=x, w=wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
    return y, [px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt]


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
    out_ch, in_ch_per_group, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # Fold the resampling filters' own footprint into the padding (same arithmetic as upsample2d/downsample2d).
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2
    pads = [px0, px1, py0, py1]
    pointwise = kh == 1 and kw == 1

    if pointwise and down > 1 and up == 1:
        x = _ufd.upfirdn2d(x=x, f=f, down=down, padding=pads, flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)

    if pointwise and up > 1 and down == 1:
        x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
        return _ufd.upfirdn2d(x=x, f=f, up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)

    if down > 1 and up == 1:
        x = _ufd.upfirdn2d(x=x, f=f, padding=pads, flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)

    if up > 1:
        # Transposed convolution wants [Cin, Cout/groups, kh, kw].
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, out_ch // groups, in_ch_per_group, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * in_ch_per_group, out_ch // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
        x = _ufd.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = _ufd.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x

    if up == 1 and down == 1 and px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
        return _conv2d_wrapper(x=x, w=w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)

    x = _ufd.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)
    x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = _ufd.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
    return x
