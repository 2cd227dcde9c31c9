"""MI355X-native StyleGAN-V hot path: hand-written gfx950 HIP kernels (csrc/, C ABI in
include/sgv_ops.h) behind the reference's own op / module interface (torch_utils/, training/)."""

__version__ = '0.1.0'
