"""Shared test helpers: golden-fixture access and comparison utilities."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Golden:
    """tests/golden/<name>.npz written by tests/golden/make_golden.py (reference outputs)."""

    def __init__(self, name):
        self.data = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.meta = json.loads(bytes(self.data['__meta__']).decode())

    def t(self, key, dtype=None, device='cpu'):
        arr = torch.from_numpy(np.array(self.data[key]))
        if dtype is not None and arr.is_floating_point():
            arr = arr.to(dtype)
        return arr.to(device)

    def has(self, key):
        return key in self.data.files

    def keys(self, prefix=''):
        return [k for k in self.data.files if k.startswith(prefix)]


def max_abs(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item() if a.numel() else 0.0


def assert_close(a, b, atol, rtol=0.0, what=''):
    a64, b64 = a.double().cpu(), b.double().cpu()
    assert a64.shape == b64.shape, f'{what}: shape {tuple(a64.shape)} vs {tuple(b64.shape)}'
    err = (a64 - b64).abs()
    tol = atol + rtol * b64.abs()
    bad = err > tol
    assert not bad.any(), f'{what}: max abs err {err.max().item():.3e} (tol {atol:g}+{rtol:g}*|ref|), {int(bad.sum())} / {bad.numel()} elements off'


def assert_bit_equal(a, b, what=''):
    assert a.shape == b.shape and a.dtype == b.dtype, f'{what}: {a.shape}/{a.dtype} vs {b.shape}/{b.dtype}'
    a, b = a.cpu().contiguous(), b.cpu().contiguous()
    # compare numerically so +0 == -0, but NaN must match NaN
    same = (a == b) | (a.isnan() & b.isnan())
    assert same.all(), f'{what}: {int((~same).sum())} / {same.numel()} elements differ, max abs {max_abs(a, b):.3e}'
