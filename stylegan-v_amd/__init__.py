"""MI355X-native StyleGAN-V hot path: hand-written gfx950 HIP kernels (csrc/, C ABI in
include/sgv_ops.h) behind the reference's own op / module interface (torch_utils/, training/)."""

__version__ = '0.1.0'


def configure_miopen(immediate=True):
    """Select MIOpen's immediate mode for the convolutions that still go to the vendor library (the 4x4 layers, weight gradients of
    layers narrower than 32 pixels, anything the hand-written 3x3 family of csrc/ does not serve, and the strict-fp32 companion of
    bench.py).  The ROCm image ships no gfx950 find-db / kernel-db: PyTorch's default
    "find" path then times every applicable solver -- including the naive direct one -- on the full-size tensors
    the first time each shape is seen (many minutes at 256^2, batch 96).  Immediate mode picks a solution
    without benchmarking (measured here: 98-118 TFLOP/s fp32 on the 3x3 layers, no start-up stall)."""
    import os
    import torch
    os.environ.setdefault('MIOPEN_FIND_MODE', '2')
    if hasattr(torch.backends, 'miopen') and hasattr(torch.backends.miopen, 'immediate'):
        torch.backends.miopen.immediate = bool(immediate)
