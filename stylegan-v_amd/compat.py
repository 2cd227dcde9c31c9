"""Drop-in installation of this package's op layer under the reference's import paths.

The reference's modules import their ops as ``src.torch_utils.ops.{upfirdn2d, bias_act, conv2d_resample,
conv2d_gradfix, fma, grid_sample_gradfix}`` (networks.py:14-16, layers.py:10-12, loss.py:13-15,
augment.py:12-16) and pickled checkpoints re-import those names when they are loaded
(persistence.py:216-227).  ``install()`` registers this package's modules under exactly those names in
``sys.modules`` so an unmodified reference checkout -- ``Generator``, ``Discriminator``, ``StyleGAN2Loss``,
``training_loop``, ``generate.py`` -- runs on the gfx950 kernels.  Call it before importing anything of the reference:

    import stylegan_v_amd.compat as compat
    compat.install()                       # src.torch_utils.ops.* -> stylegan_v_amd.torch_utils.ops.*
    from training.networks import Generator   # reference code, native ops underneath
"""

import importlib
import sys
import types

_OPS = ('upfirdn2d', 'bias_act', 'conv2d_resample', 'conv2d_gradfix', 'fma', 'grid_sample_gradfix')


def install(prefixes=('src.torch_utils.ops', 'torch_utils.ops'), also_custom_ops=True):
    """Alias the op modules (and ``custom_ops``) under every given package prefix.  Returns the mapping."""
    mapping = {}
    for prefix in prefixes:
        parts = prefix.split('.')
        for depth in range(1, len(parts) + 1):  # make sure parent packages exist without importing the reference's
            name = '.'.join(parts[:depth])
            if name not in sys.modules:
                try:
                    importlib.import_module(name)
                except Exception:
                    pkg = types.ModuleType(name)
                    pkg.__path__ = []
                    sys.modules[name] = pkg
        for op in _OPS:
            mod = importlib.import_module(f'stylegan_v_amd.torch_utils.ops.{op}')
            sys.modules[f'{prefix}.{op}'] = mod
            setattr(sys.modules[prefix], op, mod)
            mapping[f'{prefix}.{op}'] = mod
        if also_custom_ops:
            parent = '.'.join(parts[:-1])
            mod = importlib.import_module('stylegan_v_amd.torch_utils.custom_ops')
            sys.modules[f'{parent}.custom_ops'] = mod
            mapping[f'{parent}.custom_ops'] = mod
    return mapping
