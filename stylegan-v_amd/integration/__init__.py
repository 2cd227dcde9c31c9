"""Reference-side bindings (INTEGRATION.md option B): the files a universome/stylegan-v maintainer drops next to
``src/torch_utils/ops/upfirdn2d.py`` / ``bias_act.py`` / ``conv2d_gradfix.py`` to put the reference's own Python op
layer on top of ``libsgv_hip.so``.  They import nothing from this package -- only ctypes, torch and the C ABI of
``include/sgv_ops.h`` -- so they can be copied out of the tree as they are; ``SGV_HIP_LIB`` overrides the library path.
Executed by tests/test_integration_stubs.py (signatures against the reference checkout on CPU, results against the
oracle on the GPU)."""
