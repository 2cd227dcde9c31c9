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
