"""ctypes front-end of oracle/sgv_oracle.c plus small numpy restatements -- TEST INFRASTRUCTURE ONLY.

`parity pinned`: the C restatement is checked against golden vectors produced by importing the
reference's own Python implementations (tests/golden/make_golden.py -> tests/golden/*.npz) in
tests/test_oracle.py.  The reference's native CUDA sources cannot be compiled in this image.

All functions take and return CPU torch tensors (so bfloat16 works without numpy support).
"""

import ctypes
import hashlib
import os
import subprocess

import numpy as np
import torch

_DIR = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_DIR, 'sgv_oracle.c')
_LIB = os.path.join(_DIR, '_build', 'libsgv_oracle.so')
_STAMP = _LIB + '.md5'
_lib = None

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}


def _digest():
    with open(_SRC, 'rb') as fh:
        return hashlib.md5(fh.read()).hexdigest()


def build(force=False):
    """gcc the restatement into oracle/_build/ (recipe: oracle/Makefile)."""
    fresh = os.path.isfile(_LIB) and os.path.isfile(_STAMP) and open(_STAMP).read().strip() == _digest()
    if force or not fresh:
        if os.path.isfile(_LIB):
            os.remove(_LIB)
        res = subprocess.run(['make', '-C', _DIR], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('oracle build failed:\n' + res.stderr)
        with open(_STAMP, 'w') as fh:
            fh.write(_digest())
    return _LIB


def _get():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        i, i64, f, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
        lib.oracle_upfirdn2d_out_size.restype = i
        lib.oracle_upfirdn2d_out_size.argtypes = [i] * 6
        lib.oracle_upfirdn2d.restype = i
        lib.oracle_upfirdn2d.argtypes = [vp, vp, vp, i] + [i] * 8 + [i, f] + [i] * 4 + [i64] * 4 + [i, i] + [i64] * 2 + [i64] * 4
        lib.oracle_bias_act.restype = i
        lib.oracle_bias_act.argtypes = [vp] * 6 + [i, i, i, f, f, f, i64, i64, i64]
        _lib = lib
    return _lib


def upfirdn2d_out_size(in_size, up, down, pad0, pad1, taps):
    return int(_get().oracle_upfirdn2d_out_size(in_size, up, down, pad0, pad1, taps))


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _pad4(p):
    if isinstance(p, int):
        return p, p, p, p
    p = tuple(p)
    return (p[0], p[0], p[1], p[1]) if len(p) == 2 else p


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """Oracle upfirdn2d on a CPU tensor x [N,C,H,W] (any dense strides) with an fp32 filter
    ([fh,fw], [taps] separable, or None).  Separable filters run as two passes with sqrt(gain) each,
    like the reference's native path (upfirdn2d.py:236-240)."""
    assert x.device.type == 'cpu' and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    f = f.detach().to(torch.float32).cpu()
    upx, upy = _pair(up)
    downx, downy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    if f.ndim == 1:
        g = float(np.sqrt(gain))
        y = _upfirdn2d_one(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip_filter, g)
        return _upfirdn2d_one(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip_filter, g)
    return _upfirdn2d_one(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, float(gain))


def _upfirdn2d_one(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain):
    lib = _get()
    n, c, h, w = x.shape
    fh, fw = f.shape
    ow = upfirdn2d_out_size(w, upx, downx, px0, px1, fw)
    oh = upfirdn2d_out_size(h, upy, downy, py0, py1, fh)
    if ow < 1 or oh < 1:
        raise RuntimeError('output must be at least 1x1')
    cl = x.stride(1) == 1 and c > 1 and x.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, memory_format=torch.channels_last if cl else torch.contiguous_format)
    sn, sc, sh, sw = x.stride()
    on, oc, osh, osw = y.stride()
    rc = lib.oracle_upfirdn2d(x.data_ptr(), f.data_ptr(), y.data_ptr(), _DT[x.dtype],
                              upx, upy, downx, downy, px0, px1, py0, py1, int(bool(flip)), float(gain),
                              w, h, c, n, sw, sh, sc, sn, fw, fh, f.stride(1), f.stride(0), osw, osh, oc, on)
    assert rc == 0
    return y


_ACT_IDX = {'linear': 1, 'relu': 2, 'lrelu': 3, 'tanh': 4, 'sigmoid': 5, 'elu': 6, 'selu': 7, 'softplus': 8, 'swish': 9}
_ACT_DEFAULTS = {'linear': (0, 1), 'relu': (0, np.sqrt(2)), 'lrelu': (0.2, np.sqrt(2)), 'tanh': (0, 1), 'sigmoid': (0, 1),
                 'elu': (0, 1), 'selu': (0, 1), 'softplus': (0, 1), 'swish': (0, np.sqrt(2))}


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, grad=0, xref=None, yref=None, dy=None):
    """Oracle bias_act kernel call (any grad order) on dense CPU tensors sharing x's layout."""
    lib = _get()
    assert x.device.type == 'cpu' and x.is_contiguous(), 'oracle.bias_act walks dense memory: pass a contiguous tensor'
    da, dg = _ACT_DEFAULTS[act]
    alpha = float(da if alpha is None else alpha)
    gain = float(dg if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    y = torch.empty_like(x)
    ptr = lambda t: t.data_ptr() if t is not None else None
    for t in (xref, yref, dy):
        assert t is None or (t.shape == x.shape and t.stride() == x.stride() and t.dtype == x.dtype)
    rc = lib.oracle_bias_act(x.data_ptr(), ptr(b), ptr(xref), ptr(yref), ptr(dy), y.data_ptr(), _DT[x.dtype], grad,
                             _ACT_IDX[act], alpha, gain, clamp, x.numel(), b.numel() if b is not None else 1,
                             x.stride(dim) if b is not None else 1)
    assert rc == 0
    return y


# ----------------------------------------------------------------------------------------------
# numpy restatements of the small fp32 formulas (python loops are fine at test sizes).


def modulated_demod_coefs(weight, styles, eps=1e-8):
    """d[n,o] = rsqrt(sum_{i,kh,kw} (W[o,i,kh,kw]*s[n,i])^2 + eps) -- networks.py:57-62, computed the
    reference's way (materialising w[N,O,I,kh,kw]) in float64, returned as float64."""
    w = weight.detach().double().numpy()
    s = styles.detach().double().numpy()
    ww = w[None] * s[:, None, :, None, None]
    return torch.from_numpy(1.0 / np.sqrt((ww ** 2).sum(axis=(2, 3, 4)) + eps))


def time_encode(periods, phases, al, ar, freqs, phase_scales, t, t_left, t_right, alpha):
    """AlignedTimeEncoder tail (motion.py:201-212) in float64 numpy.  periods include tanh()+1."""
    P, Ph = periods.double().numpy(), phases.double().numpy()
    fr, ps = freqs.double().numpy().reshape(1, -1), phase_scales.double().numpy().reshape(1, -1)
    a = alpha.double().numpy().reshape(-1, 1)

    def pos(tau):
        raw = fr * P * tau.double().numpy().reshape(-1, 1) + Ph * ps
        return np.concatenate([np.sin(raw), np.cos(raw)], axis=1)

    rem = pos(t_left) * (1 - a) + pos(t_right) * a
    add = al.double().numpy() * (1 - a) + ar.double().numpy() * a
    return torch.from_numpy(pos(t) - rem + add)


# ----------------------------------------------------------------------------------------------
# Differentiable float64 restatements in plain torch ops (comparators of the gradient / second-order tests: autograd differentiates the
# definition itself, nothing of the product package is involved).


def demod_coefs_torch(weight, styles, eps=1e-8):
    """networks.py:57-62 the reference's way -- w[N,O,I,kh,kw] = W * s, d = rsqrt(sum w^2 + eps) -- as a differentiable torch expression."""
    w = weight.unsqueeze(0) * styles.reshape(styles.shape[0], 1, -1, 1, 1)
    return (w.square().sum(dim=[2, 3, 4]) + eps).rsqrt()


def dense(x, weight, bias=None, weight_gain=1.0, bias_gain=1.0, act='linear', normalize=False, act_gain=None, alpha=0.2):
    """FullyConnectedLayer.forward (layers.py:126-137: w = weight * weight_gain, b = bias * bias_gain, x @ w.t() + b, bias_act) behind the optional
    normalize_2nd_moment of the mapping network (layers.py:22-25); activations 'linear' and 'lrelu' (bias_act.py:26,29: lrelu slope 0.2, gain sqrt 2)."""
    if normalize:
        x = x * (x.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
    y = x.matmul((weight * weight_gain).t())
    if bias is not None:
        y = y + (bias * bias_gain).unsqueeze(0)
    assert act in ('linear', 'lrelu')
    gain = act_gain if act_gain is not None else (1.0 if act == 'linear' else float(np.sqrt(2.0)))
    if act == 'lrelu':
        y = torch.nn.functional.leaky_relu(y, alpha)
    return y * gain if gain != 1 else y


def affine_resample(x, theta, out_hw):
    """The geometric resampling of AugmentPipe (augment.py:295-297): affine_grid(align_corners=False) + bilinear grid_sample with zero padding, written
    out as four gathers (torch's own grid_sample has no second derivative -- the reason the reference carries grid_sample_gradfix.py): differentiable to
    any order in x.  Sampling positions per the ATen definition: output pixel centre (2j + 1) / Wo - 1 in [-1, 1], source index ((g + 1) * W - 1) / 2."""
    n, c, h, w = x.shape
    ho, wo = out_hw
    xs = (2 * torch.arange(wo, dtype=x.dtype, device=x.device) + 1) / wo - 1
    ys = (2 * torch.arange(ho, dtype=x.dtype, device=x.device) + 1) / ho - 1
    base = torch.stack([xs.reshape(1, wo).expand(ho, wo), ys.reshape(ho, 1).expand(ho, wo), torch.ones([ho, wo], dtype=x.dtype, device=x.device)], dim=-1)  # [Ho, Wo, 3]
    g = torch.einsum('hwk,nik->nhwi', base, theta.to(x.dtype))      # [N, Ho, Wo, 2] = (gx, gy)
    ix = ((g[..., 0] + 1) * w - 1) / 2
    iy = ((g[..., 1] + 1) * h - 1) / 2
    x0, y0 = torch.floor(ix), torch.floor(iy)
    fx, fy = ix - x0, iy - y0
    flat = x.reshape(n, c, h * w)
    out = 0
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xc, yc = x0 + dx, y0 + dy
            ok = ((xc >= 0) & (xc <= w - 1) & (yc >= 0) & (yc <= h - 1)).to(x.dtype)
            idx = (yc.clamp(0, h - 1) * w + xc.clamp(0, w - 1)).long().reshape(n, 1, ho * wo).expand(n, c, ho * wo)
            out = out + torch.gather(flat, 2, idx).reshape(n, c, ho, wo) * (wx * wy * ok).unsqueeze(1)
    return out


def ada_geometric(x, theta, taps, margin):
    """The geometric execution block of AugmentPipe.forward (src/training/augment.py:284-303) in float64, given what the block receives: the image batch
    x [N,C,H,W], the margin (mx0, mx1, my0, my1) of :282, the 12 `Hz_geom` taps and theta = G_inv[:, :2, :] as passed to `affine_grid` at :299 (the map
    from the (2 (H + 6)) x (2 (W + 6)) resampled image to the up-sampled padded image, normalised coordinates).  Steps, each by its definition:
      reflect pad          torch.nn.functional.pad(mode='reflect')                                            :284
      2x up-sampling       upfirdn2d.upsample2d(f=Hz_geom, up=2): padding [6, 5], gain 4 (upfirdn2d.py:308-343)    :288
      resampling           affine_grid(align_corners=False) + bilinear grid_sample, zeros outside            :299-300
      2x down-sampling     upfirdn2d.downsample2d(f=Hz_geom, down=2, padding=-6, flip_filter=True)             :303
    The FIR steps go through this module's C restatement of upfirdn2d_kernel_large (`upfirdn2d` above), the resampling through `affine_resample`."""
    x = torch.as_tensor(np.asarray(x, dtype=np.float64))
    f = torch.as_tensor(np.asarray(taps, dtype=np.float32))       # the reference's filters are fp32 whatever the image format (upfirdn2d.cpp:20-21); 1-D: two passes
    assert f.ndim == 1 and f.shape[0] % 4 == 0
    fw = f.shape[0]
    pad = fw // 4
    mx0, mx1, my0, my1 = (int(m) for m in margin)
    n, c, h, w = x.shape
    xp = torch.nn.functional.pad(x, [mx0, mx1, my0, my1], mode='reflect')
    p0, p1 = (fw + 2 - 1) // 2, (fw - 2) // 2                     # upsample2d's padding for up = 2 (upfirdn2d.py:336-341)
    up = upfirdn2d(xp, f, up=2, padding=[p0, p1, p0, p1], gain=4.0)
    hi = affine_resample(up, torch.as_tensor(np.asarray(theta, dtype=np.float64)), ((h + 2 * pad) * 2, (w + 2 * pad) * 2))
    q0, q1 = -2 * pad + (fw - 2 + 1) // 2, -2 * pad + (fw - 2) // 2  # downsample2d's padding for down = 2 plus the block's -2 * Hz_pad (upfirdn2d.py:375-380)
    return upfirdn2d(hi, f, down=2, padding=[q0, q1, q0, q1], flip_filter=True).numpy()


# ----------------------------------------------------------------------------------------------
# 3x3 convolutions.  The reference leaves them to ATen / cuDNN (call sites: conv2d_gradfix.py:35-43 `conv2d` /
# `conv_transpose2d`, :100-118 data gradient, :140-170 weight gradient; conv2d_resample.py:113-137 for the stride-2 forms) --
# third-party arithmetic pinned by the reference only as "pytorch 1.7.1 / 1.9" (environment.yaml:9-11), no tests or golden
# vectors at that boundary: parity there is UNPINNED against the reference itself.  The restatement below is the textbook
# definition in float64 (tap loop + channel contraction); tests/test_oracle.py checks it against torch's CPU float64
# `F.conv2d` / `F.conv_transpose2d`, the same ATen entry points the reference calls.

def conv3x3(x, w, stride=1, transposed=False):
    """float64 direct 3x3 convolution on NCHW.
    stride 1: padding 1.  `transposed=False`: y[n,m,Y,X] = sum w[m,k,ky,kx] x[n,k,s*Y+ky-p,s*X+kx-p];  stride 2: padding 0.
    `transposed=True` (w is [K, M, 3, 3], conv_transpose2d semantics): y[n,m,s*Y+ky-p,s*X+kx-p] += w[k,m,ky,kx] x[n,k,Y,X]."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    assert stride in (1, 2) and x.ndim == 4 and w.shape[2:] == (3, 3)
    pad = 1 if stride == 1 else 0
    n, k, h, wd = x.shape
    if not transposed:
        assert w.shape[1] == k
        ho, wo = (h + 2 * pad - 3) // stride + 1, (wd + 2 * pad - 3) // stride + 1
        xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
        y = np.zeros([n, w.shape[0], ho, wo])
        for ky in range(3):
            for kx in range(3):
                y += np.einsum('mk,nkhw->nmhw', w[:, :, ky, kx], xp[:, :, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride])
        return y
    assert w.shape[0] == k
    ho, wo = (h - 1) * stride - 2 * pad + 3, (wd - 1) * stride - 2 * pad + 3
    full = np.zeros([n, w.shape[1], (h - 1) * stride + 3, (wd - 1) * stride + 3])
    for ky in range(3):
        for kx in range(3):
            full[:, :, ky:ky + stride * (h - 1) + 1:stride, kx:kx + stride * (wd - 1) + 1:stride] += np.einsum('km,nkhw->nmhw', w[:, :, ky, kx], x)
    return full[:, :, pad:pad + ho, pad:pad + wo]


def conv3x3_weight_grad(dy, x, stride=1, transposed=False):
    """float64 gradient of <dy, conv3x3(x, w)> with respect to w (same layout as w)."""
    dy = np.asarray(dy, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    pad = 1 if stride == 1 else 0
    if transposed:   # y[.., s*Y+ky-p, ..] += w[k,m] x[k,Y]  ->  dw[k,m,ky,kx] = sum x[n,k,Y,X] dy[n,m,s*Y+ky-p,s*X+kx-p]
        small, big = x, dy
    else:            # dw[m,k,ky,kx] = sum dy[n,m,Y,X] x[n,k,s*Y+ky-p,s*X+kx-p]
        small, big = dy, x
    hs, ws = small.shape[2:]
    bp = np.pad(big, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    dw = np.zeros([small.shape[1], big.shape[1], 3, 3])
    for ky in range(3):
        for kx in range(3):
            dw[:, :, ky, kx] = np.einsum('nshw,nbhw->sb', small, bp[:, :, ky:ky + stride * (hs - 1) + 1:stride, kx:kx + stride * (ws - 1) + 1:stride])
    return dw
