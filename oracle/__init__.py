"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/sgv_oracle.c header).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package.
"""
from .oracle import (build, upfirdn2d, bias_act, upfirdn2d_out_size, modulated_demod_coefs, time_encode, conv3x3, conv3x3_weight_grad,  # noqa: F401
                     demod_coefs_torch, dense, affine_resample, ada_geometric)
