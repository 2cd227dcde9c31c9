"""Affine bilinear resampling of an image batch: ``affine_resample(x, theta, (Ho, Wo))`` ==
``grid_sample(x, affine_grid(theta, [N, C, Ho, Wo], align_corners=False), bilinear, zeros, align_corners=False)``.

This is the geometric execution step of the ADA augmentation pipeline (reference: src/training/augment.py:297-300 through
src/torch_utils/ops/grid_sample_gradfix.py).  On the GPU it is one gather kernel that evaluates the sampling grid in registers
(csrc/resample.hip) instead of materialising it; the gradient w.r.t. the image is the adjoint scatter kernel, and since both maps are
linear in the image one autograd node serves either direction -- differentiating it gives the other -- so the R1 penalty (a gradient of
a gradient through augmented real images, loss.py:144-164) works.  ``theta`` (built from random augmentation parameters) gets no
gradient.  CPU / non-fp32 tensors take the two-op formulation through ``grid_sample_gradfix``.
"""

import os

import torch

from .. import custom_ops
from . import grid_sample_gradfix

enabled = True
one_kernel_backward = os.environ.get('SGV_ADA_ADJOINT', '1') != '0'      # lab switch: 0 = differentiated calls of the ADA block run the four-pass composition (rounds 4-5)


def affine_resample_ref(x, theta, out_hw):
    grid = torch.nn.functional.affine_grid(theta, [x.shape[0], x.shape[1], out_hw[0], out_hw[1]], align_corners=False)
    return grid_sample_gradfix.grid_sample(x, grid)


class _AffineMap(torch.autograd.Function):
    """``src_hw is None``: y = S(t), t the image [N,C,H,W] -> [N,C,Ho,Wo].  ``src_hw = (H, W)``: y = S^T(t), t [N,C,Ho,Wo] -> [N,C,H,W]."""

    @staticmethod
    def forward(ctx, t, theta, out_hw, src_hw):
        lib = custom_ops.get_native()
        t, th = t.contiguous(), theta.detach().contiguous().float()
        n, c = t.shape[:2]
        adjoint = src_hw is not None
        h, w = src_hw if adjoint else t.shape[2:]
        ho, wo = out_hw
        dst = torch.zeros([n, c, h, w], dtype=torch.float32, device=t.device) if adjoint else torch.empty([n, c, ho, wo], dtype=torch.float32, device=t.device)
        with custom_ops.device_guard(t):
            custom_ops.check(lib.sgv_affine_resample(t.data_ptr(), dst.data_ptr(), th.data_ptr(), n, c, h, w, ho, wo, int(adjoint), custom_ops.raw_stream(t)), lib)
        ctx.adjoint, ctx.out_hw, ctx.img_hw = adjoint, (ho, wo), (h, w)
        ctx.save_for_backward(th)
        return dst

    @staticmethod
    def backward(ctx, g):
        (th,) = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise RuntimeError('affine_resample: no gradient w.r.t. theta (augmentation parameters are constants)')
        d_t = _AffineMap.apply(g, th, ctx.out_hw, None if ctx.adjoint else ctx.img_hw) if ctx.needs_input_grad[0] else None
        return d_t, None, None, None


def affine_resample(x, theta, out_hw):
    """x [N,C,H,W], theta [N,2,3] (normalised coordinates, as ``affine_grid``), out_hw = (Ho, Wo)."""
    out_hw = (int(out_hw[0]), int(out_hw[1]))
    if enabled and x.is_cuda and x.dtype == torch.float32 and x.ndim == 4 and not theta.requires_grad:
        return _AffineMap.apply(x, theta, out_hw, None)
    return affine_resample_ref(x, theta.to(x.dtype), out_hw)


def ada_geometric_ref(x, theta, f, margin):
    """The four-pass composition of AugmentPipe's geometric execution (augment.py:270-300) the fused kernel replaces; ``theta`` is the map the resampling
    step gets, ``margin`` = (mx0, mx1, my0, my1), ``f`` the 12-tap filter."""
    from . import upfirdn2d
    n, c, h, w = x.shape
    pad = f.shape[0] // 4
    mx0, mx1, my0, my1 = margin
    y = torch.nn.functional.pad(x, [mx0, mx1, my0, my1], mode='reflect')
    y = upfirdn2d.upsample2d(y, f, up=2)
    y = affine_resample(y, theta, ((h + pad * 2) * 2, (w + pad * 2) * 2))
    return upfirdn2d.downsample2d(y, f, down=2, padding=-pad * 2, flip_filter=True)


def ada_geometric_fused_ok(x, f):
    """One-kernel forms (csrc/resample.hip `ada_geometric_forward_kernel` / `ada_geometric_adjoint_kernel`): fp32 CUDA images, the 12-tap filter."""
    return (enabled and x.is_cuda and x.dtype == torch.float32 and x.ndim == 4 and f.ndim == 1 and f.shape[0] == 12 and min(x.shape[2:]) >= 2
            and x.shape[0] <= 65535)


def _launch_geometric(t, theta, taps, margin, adjoint):
    import ctypes
    lib = custom_ops.get_native()
    tc = t.contiguous()
    n, c, h, w = tc.shape
    out = torch.empty_like(tc)
    ctaps = (ctypes.c_float * 12)(*taps)
    with custom_ops.device_guard(tc):
        fn = lib.sgv_ada_geometric_adjoint if adjoint else lib.sgv_ada_geometric
        custom_ops.check(fn(tc.data_ptr(), out.data_ptr(), theta.data_ptr(), ctypes.addressof(ctaps), n, c, h, w, *margin, custom_ops.raw_stream(tc)), lib)
    return out


class _AdaGeometric(torch.autograd.Function):
    """y = A t (`adjoint` False: the block of augment.py:284-303 as one kernel) or A^T t (`adjoint` True: its backward as one kernel).  A is linear in the image,
    so the derivative of either direction is the other: the node differentiates to any order -- the generator's phase (loss.py:91-110: one backward through
    augmented fakes) and the R1 penalty (loss.py:144-164: a gradient of a gradient through augmented reals) run these two kernels and nothing else."""

    @staticmethod
    def forward(ctx, t, theta, taps, margin, adjoint):
        ctx.args = (theta, taps, margin, adjoint)
        return _launch_geometric(t, theta, taps, margin, adjoint)

    @staticmethod
    def backward(ctx, g):
        theta, taps, margin, adjoint = ctx.args
        return (_AdaGeometric.apply(g, theta, taps, margin, not adjoint) if ctx.needs_input_grad[0] else None), None, None, None, None


def ada_geometric(x, theta, f, margin, f_host=None):
    """y = down2(resample(up2(reflect_pad(x)))) -- ONE launch when ``ada_geometric_fused_ok``, and one (plus an empty second one, see sgv_ops.h) per backward
    pass of any order.  ``f_host``: the filter's taps as a list of floats (they travel as launch arguments; read back once otherwise)."""
    margin = tuple(int(m) for m in margin)
    if not ada_geometric_fused_ok(x, f) or theta.requires_grad:
        return ada_geometric_ref(x, theta, f, margin)
    taps = tuple(f_host if f_host is not None else f.detach().cpu().tolist())
    th = theta.detach().contiguous().float()
    if x.requires_grad and torch.is_grad_enabled():
        if not one_kernel_backward:
            return ada_geometric_ref(x, theta, f, margin)
        return _AdaGeometric.apply(x, th, taps, margin, False)
    return _launch_geometric(x.detach(), th, taps, margin, False)
