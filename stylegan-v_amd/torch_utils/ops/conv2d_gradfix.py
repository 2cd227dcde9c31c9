"""conv2d / conv_transpose2d entry points of the op layer.

Boundary names of the reference's ``src/torch_utils/ops/conv2d_gradfix.py`` (``conv2d`` :35,
``conv_transpose2d`` :40, ``no_weight_gradients`` :26, module globals ``enabled`` :22 and
``weight_gradients_disabled`` :23), which ``loss.py``, ``training_loop.py`` and ``augment.py`` import
by name.  The reference's custom autograd path only ever activates on torch 1.7-1.10
(conv2d_gradfix.py:53) and calls cuDNN-only ATen ops; on current PyTorch the stock convolutions
already support arbitrary-order gradients, so both functions forward to ``torch.nn.functional``
(MIOpen on ROCm).  ``no_weight_gradients()`` keeps its flag semantics so callers can still query it.
"""

import contextlib

import torch

enabled = False                    # kept for API compatibility; has no effect
weight_gradients_disabled = False  # set inside no_weight_gradients()


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    previous = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = previous


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                      dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)
