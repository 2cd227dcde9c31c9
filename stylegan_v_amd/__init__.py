"""Import alias: ``stylegan_v_amd`` -> the package that lives in ``stylegan-v_amd/``.

The product directory is named after the upstream repository (``stylegan-v_amd``), which is not a
valid Python identifier.  This stub makes it importable: it points its ``__path__`` at that
directory and executes its ``__init__.py`` in this module's namespace, so
``import stylegan_v_amd.torch_utils.ops.upfirdn2d`` resolves to
``stylegan-v_amd/torch_utils/ops/upfirdn2d.py``.  No code lives here.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'stylegan-v_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _fh:
    exec(compile(_fh.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _fh
