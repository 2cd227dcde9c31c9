"""Small host-side helpers the op layer and the modules share.

Boundary names kept from the reference's ``src/torch_utils/misc.py`` (``assert_shape`` :77,
``profiled_function`` :99, ``suppress_tracer_warnings`` :66, ``nan_to_num`` :46,
``named_params_and_buffers`` :145, ``copy_params_and_buffers`` :149, ``ddp_sync`` :167,
``check_ddp_consistency`` :179) because modules and drivers above the ops import them by name.
"""

import contextlib
import functools
import re
import warnings

import torch

nan_to_num = torch.nan_to_num


def nan_to_num_list_(tensors, nan=0.0, posinf=None, neginf=None):
    """In-place ``nan_to_num`` over a list of tensors: the per-parameter gradient loop of training_loop.py:384-386.  Dense fp32 GPU tensors
    go through ONE native launch per 96 tensors (csrc/multi_tensor.hip); anything else takes torch's op, tensor by tensor."""
    native, rest = [], []
    for t in tensors:
        (native if (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) else rest).append(t)
    for t in rest:
        torch.nan_to_num(t, nan=nan, posinf=posinf, neginf=neginf, out=t)
    if native:
        import ctypes
        from . import custom_ops
        lib = custom_ops.get_native()
        fin = torch.finfo(torch.float32)
        ptrs = (ctypes.c_void_p * len(native))(*[t.data_ptr() for t in native])
        sizes = (ctypes.c_int64 * len(native))(*[t.numel() for t in native])
        with custom_ops.device_guard(native[0]):
            custom_ops.check(lib.sgv_multi_nan_to_num_f32(ptrs, sizes, len(native), float(nan), float(fin.max if posinf is None else posinf),
                                                          float(fin.min if neginf is None else neginf), custom_ops.raw_stream(native[0])), lib)
    return tensors


class suppress_tracer_warnings(warnings.catch_warnings):
    """``with`` block that silences torch.jit tracer warnings (shape values used as constants)."""

    def __enter__(self):
        super().__enter__()
        warnings.simplefilter('ignore', category=torch.jit.TracerWarning)
        return self


def assert_shape(tensor, ref_shape):
    """Raise AssertionError unless ``tensor.shape`` matches ``ref_shape`` (``None`` = any size)."""
    shape = tuple(tensor.shape)
    if len(shape) != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {len(shape)}, expected {len(ref_shape)} for tensor of size {list(shape)}')
    for dim, (got, want) in enumerate(zip(shape, ref_shape)):
        if want is not None and int(got) != int(want):
            raise AssertionError(f'Wrong size for dimension {dim}: got {got}, expected {want} for tensor of size {list(shape)}')


def profiled_function(fn):
    """Decorator: run ``fn`` inside a ``record_function`` range named after it (shows up in torch.profiler / roctx)."""

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)

    return wrapped


def named_params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.named_parameters()) + list(module.named_buffers())


def params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.parameters()) + list(module.buffers())


def copy_params_and_buffers(src_module, dst_module, require_all=False):
    """Copy same-named parameters/buffers from ``src_module`` into ``dst_module`` (requires_grad preserved)."""
    source = dict(named_params_and_buffers(src_module))
    for name, tensor in named_params_and_buffers(dst_module):
        assert name in source or not require_all, f'{name} missing in source module'
        if name in source:
            tensor.copy_(source[name].detach()).requires_grad_(tensor.requires_grad)


@contextlib.contextmanager
def ddp_sync(module, sync):
    """Run the block with DDP gradient all-reduce enabled (``sync``) or suppressed via ``no_sync()``.

    This is what gates RCCL traffic per phase: only the last accumulation round of the module being
    optimised synchronises (loss.py:45,52,69 in the reference)."""
    assert isinstance(module, torch.nn.Module)
    if sync or not isinstance(module, torch.nn.parallel.DistributedDataParallel):
        yield
    else:
        with module.no_sync():
            yield


def check_ddp_consistency(module, ignore_regex=None):
    """Assert every parameter/buffer equals rank 0's copy (one broadcast per tensor)."""
    assert isinstance(module, torch.nn.Module)
    for name, tensor in named_params_and_buffers(module):
        fullname = type(module).__name__ + '.' + name
        if ignore_regex is not None and re.fullmatch(ignore_regex, fullname):
            continue
        mine = tensor.detach()
        theirs = mine.clone()
        torch.distributed.broadcast(tensor=theirs, src=0)
        assert (nan_to_num(mine) == nan_to_num(theirs)).all(), f'{fullname} is not DDP consistent'
