"""Reference-side replacement of the geometric execution block of ``AugmentPipe.forward`` (src/training/augment.py:284-303: reflect pad -> ``upfirdn2d.upsample2d`` ->
``affine_grid`` + ``grid_sample_gradfix.grid_sample`` -> ``upfirdn2d.downsample2d``) by the two C-ABI calls ``sgv_ada_geometric`` / ``sgv_ada_geometric_adjoint``
(include/sgv_ops.h).  No import from this package: ctypes on the shared library, torch for memory and the stream.

Reference-side edit: keep augment.py:272-283 (margin) and :285-296 (the matrix bookkeeping that ends in ``G_inv``), then instead of :284, :288, :299-300, :303

    images = sgv_augment.ada_geometric(images, G_inv[:, :2, :], self.Hz_geom, (mx0, mx1, my0, my1))

The block is linear in the image, so ONE autograd function serves both directions: its backward applies the other direction, which differentiates again -- first
order for the generator's phase (loss.py:91-110), second order for the R1 penalty through augmented reals (loss.py:144-164).
"""
import ctypes
import os

import torch

_LIB_PATH = os.environ.get('SGV_HIP_LIB') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc', 'libsgv_hip.so')
_lib = None
_i32, _vp = ctypes.c_int32, ctypes.c_void_p


def _get():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_LIB_PATH)     # OSError if the library is missing: no silent fallback
        lib.sgv_last_error.restype = ctypes.c_char_p
        for fn in (lib.sgv_ada_geometric, lib.sgv_ada_geometric_adjoint):
            fn.argtypes = [_vp, _vp, _vp, _vp] + [_i32] * 8 + [_vp]
        _lib = lib
    return _lib


def _run(t, theta, taps, margin, adjoint):
    lib = _get()
    t = t.contiguous()
    n, c, h, w = t.shape
    out = torch.empty_like(t)
    ctaps = (ctypes.c_float * 12)(*taps)                           # HOST pointer: the taps travel as launch arguments
    fn = lib.sgv_ada_geometric_adjoint if adjoint else lib.sgv_ada_geometric
    with torch.cuda.device(t.device):
        rc = fn(t.data_ptr(), out.data_ptr(), theta.data_ptr(), ctypes.addressof(ctaps), n, c, h, w, *margin, _vp(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(lib.sgv_last_error().decode(errors='replace'))
    return out


class _Block(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, theta, taps, margin, adjoint):
        ctx.args = (theta, taps, margin, adjoint)
        return _run(t, theta, taps, margin, adjoint)

    @staticmethod
    def backward(ctx, g):
        theta, taps, margin, adjoint = ctx.args
        return _Block.apply(g, theta, taps, margin, not adjoint), None, None, None, None


def ada_geometric(images, theta, Hz_geom, margin):
    """images [N,C,H,W] fp32 on the GPU; theta [N,2,3] = ``G_inv[:, :2, :]`` of augment.py:299; Hz_geom: the 12-tap filter buffer; margin = (mx0, mx1, my0, my1) of :282."""
    assert images.is_cuda and images.dtype == torch.float32 and Hz_geom.numel() == 12
    taps = tuple(float(v) for v in Hz_geom.detach().cpu().tolist())        # (a constant of the pipe: read once, e.g. in __init__)
    return _Block.apply(images, theta.detach().float().contiguous(), taps, tuple(int(m) for m in margin), False)
