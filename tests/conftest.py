import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MIOPEN_FIND_MODE', '2')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) if somebody runs the whole suite on a box without a GPU."""
    import torch
    import stylegan_v_amd
    stylegan_v_amd.configure_miopen()
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _selftest_before_the_first_gpu_test():
    """The load-time self-test of the asm-load convolution kernels (ops/selftest.py) runs on the first native convolution of a process; tests that count
    launches or pin kernel variants would see its eight launches inside whichever of them comes first.  Run it here, once, ahead of every test."""
    import torch
    if torch.cuda.is_available():
        from stylegan_v_amd.torch_utils import custom_ops
        from stylegan_v_amd.torch_utils.ops import conv2d_gradfix
        if custom_ops.is_built():
            conv2d_gradfix._selftest(torch.zeros(1, device='cuda'))
    yield
