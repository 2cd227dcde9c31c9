"""Magnitude bounds for the block-scaled fp16 split (`terms = 4` of the 3x3 family and the tiled GEMM; csrc/sgv_split.h).

The reference multiplies in strict fp32 (src/training/training_loop.py:129,141-142: ``allow_tf32 = False``).  gfx950 has no fp32-rate matrix
path, so the fp32-grade arithmetic of this build splits every operand into two fp16 terms of the tensor scaled by a power of two -- and that
scale needs an upper bound of the tensor's largest magnitude, in device memory (nothing is read back to the host: the launch stays
hipGraph-capturable).  ``bound(t)`` returns such a bound as a 1-element fp32 tensor:

* a tensor that was produced by a kernel of this library which left a bound behind carries it already (``attach``);
* otherwise one streaming pass (``sgv_absmax``) computes max |t|;
* the result is cached ON the tensor object (an attribute dies with the object: no stale entries after the allocator reuses the address) and is
  dropped when the tensor's version counter moves.  Kernels of this library that write into an existing tensor through its raw pointer
  (``accumulate`` stores, the in-place residual add) do not move that counter: they call ``invalidate``.
* a replayed hipGraph writes tensors (the parameters of a captured Adam step, every captured activation) without moving any version counter and without
  running this module at all: ``graph_replayed()`` -- called by the step after every replay -- advances an epoch that is part of every cache entry,
  so no bound computed before a replay is trusted after it (the eager Greg / Dreg phases of a captured schedule used to multiply with a
  weight bound taken before the latest replayed update; only the split's 2x headroom covered it).
"""

import torch

from .. import custom_ops

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_ATTR = '_sgv_amax'
_TAPS_ATTR = '_sgv_tap_sum'
_epoch = 0         # advanced by graph_replayed(): cache entries of an older epoch are stale


def graph_replayed():
    """A hipGraph was replayed: tensors it writes changed behind every version counter.  Drops every cached bound (lazily: by epoch)."""
    global _epoch
    _epoch += 1


_ARENA = 4096
_arenas = {}       # device index -> [zeroed fp32 tensor, slots handed out]
SINK_SLOTS = 4096              # SGV_AMAX_SINK_SLOTS of include/sgv_ops.h
_SINK_BLOCK = SINK_SLOTS + 64  # floats per sink block (1 + slots, padded to a 256-byte multiple)
_SINKS_PER_ARENA = 64
_sink_arenas = {}  # device index -> [zeroed fp32 tensor, blocks handed out]


# While a hipGraph is being captured, slots and sinks come from arenas that belong to THAT capture: the arena's `zeros` is a fill node of the graph in front of
# every kernel that uses a slot carved from it (capture order = replay order), so every replay starts from zero -- one fill per 4,096 bounds / 64 sinks instead of
# one per bound (round 5: 338 fill launches per captured 8-video step, 1.6 ms of 50, profiles/r05_c2_kernels_b8_graphs.txt).  `capture_scope()` brackets a capture
# (training/train_step.py); outside such a scope a capture falls back to one `zeros` per slot.
_capture = None     # {'slots': [tensor, used], 'sinks': [tensor, used]} while a bracketed capture is running


class capture_scope:
    def __enter__(self):
        global _capture
        self._saved, _capture = _capture, dict(slots=None, sinks=None, zeros=None)
        return self

    def __exit__(self, *exc):
        global _capture
        _capture = self._saved
        return False


def _capture_carve(kind, device, block, per_arena):
    a = _capture[kind]
    if a is None or a[1] >= per_arena or a[0].device != device:
        a = _capture[kind] = [torch.zeros([per_arena * block], dtype=torch.float32, device=device), 0]
    a[1] += 1
    return a[0][(a[1] - 1) * block:a[1] * block]


def zero_sink(device):
    """A zeroed block for the library's bound side output (sgv_amax_sink): [0] receives the bound, [1 .. SINK_SLOTS] the producer's partial maxima.
    Carved from a zeroed arena like `zero_slot` (one 1-MiB fill per 64 sinks)."""
    if torch.cuda.is_current_stream_capturing():
        if _capture is not None:
            return _capture_carve('sinks', device, _SINK_BLOCK, _SINKS_PER_ARENA)
        return torch.zeros([_SINK_BLOCK], dtype=torch.float32, device=device)
    a = _sink_arenas.get(device.index)
    if a is None or a[1] >= _SINKS_PER_ARENA:
        a = _sink_arenas[device.index] = [torch.zeros([_SINKS_PER_ARENA * _SINK_BLOCK], dtype=torch.float32, device=device), 0]
    a[1] += 1
    return a[0][(a[1] - 1) * _SINK_BLOCK:a[1] * _SINK_BLOCK]


_ZBLOCK = 1 << 20          # floats per arena of small zeroed work buffers (4 MiB)
_zero_arenas = {}          # device index -> [zeroed fp32 tensor, floats handed out]


def zeros(shape, device):
    """An fp32 tensor of zeros for a kernel's small accumulation target (per-plane sums, dot products, bias-gradient partials): carved from a zeroed arena, 256-byte
    aligned, never handed out twice -- one fill launch per arena instead of one per buffer (round 6: ~130 fill launches per training iteration).  While a hipGraph
    is being captured the arena belongs to the capture, like `zero_slot`'s.  Large requests (> 1/8 arena) get their own `torch.zeros`."""
    numel = 1
    for d in shape:
        numel *= int(d)
    device = torch.device(device)
    if device.type != 'cuda' or numel == 0 or numel > _ZBLOCK // 8:
        return torch.zeros(list(shape), dtype=torch.float32, device=device)
    need = (numel + 63) // 64 * 64
    if torch.cuda.is_current_stream_capturing():
        if _capture is None:
            return torch.zeros(list(shape), dtype=torch.float32, device=device)
        a = _capture.get('zeros')
        if a is None or a[1] + need > _ZBLOCK or a[0].device != device:
            a = _capture['zeros'] = [torch.zeros([_ZBLOCK], dtype=torch.float32, device=device), 0]
    else:
        a = _zero_arenas.get(device.index)
        if a is None or a[1] + need > _ZBLOCK:
            a = _zero_arenas[device.index] = [torch.zeros([_ZBLOCK], dtype=torch.float32, device=device), 0]
    off = a[1]
    a[1] += need
    return a[0][off:off + numel].view(list(shape))


def zero_slot(device):
    """A 1-element fp32 device tensor that holds 0.0f: what the bound kernels fold max |v| into (atomicMax of the bit pattern).  Slots are carved from a
    zeroed arena -- one fill launch per 4,096 bounds instead of one clearing command per bound (~600 per training step) -- and never handed out twice;
    a view keeps its arena alive.  While a hipGraph is being captured the arena belongs to the capture (`capture_scope`: its fill is a node of THAT graph, so
    every replay starts from zero; an arena cleared outside the graph would only ever grow)."""
    if torch.cuda.is_current_stream_capturing():
        if _capture is not None:
            return _capture_carve('slots', device, 1, _ARENA)
        return torch.zeros([1], dtype=torch.float32, device=device)
    a = _arenas.get(device.index)
    if a is None or a[1] >= _ARENA:
        a = _arenas[device.index] = [torch.zeros([_ARENA], dtype=torch.float32, device=device), 0]
    a[1] += 1
    return a[0][a[1] - 1:a[1]]


def bound(t):
    """[1] fp32 device tensor >= max |t| for a dense CUDA tensor t (fp32 / fp16 / bf16)."""
    cached = getattr(t, _ATTR, None)
    if cached is not None and cached[0] == t._version and cached[1] == t.data_ptr() and cached[3] == _epoch:
        return cached[2]
    assert t.is_cuda and t.dtype in _DT
    if _trace is not None:
        _trace_pass(t)
    tc = t if t.is_contiguous() else t.contiguous()
    out = zero_slot(t.device)
    lib = custom_ops.get_native()
    with custom_ops.device_guard(tc):
        custom_ops.check(lib.sgv_absmax(tc.data_ptr(), tc.numel(), _DT[t.dtype], out.data_ptr(), 1, custom_ops.raw_stream(tc)), lib)
    attach(t, out)
    return out


# SGV_AMAX_TRACE=1: which call sites still need a separate pass (tensor shape x caller), printed at exit -- the work list for further producer fusion
_trace = {} if __import__('os').environ.get('SGV_AMAX_TRACE') == '1' else None


def _trace_pass(t):
    import sys
    f = sys._getframe(2)
    chain = []
    while f is not None and len(chain) < 3:
        chain.append(f'{f.f_code.co_name}:{f.f_lineno}')
        f = f.f_back
    key = (tuple(t.shape), ' < '.join(chain))
    e = _trace.setdefault(key, [0, 0])
    e[0] += 1
    e[1] += t.numel() * t.element_size()


if _trace is not None:
    import atexit

    def _dump():
        import sys
        rows = sorted(_trace.items(), key=lambda kv: -kv[1][1])
        print('[amax trace] separate bound passes: %d launches, %.1f GB' % (sum(v[0] for _, v in rows), sum(v[1] for _, v in rows) / 1e9), file=sys.stderr)
        for (shape, who), (n, nbytes) in rows[:40]:
            print('  %6d x %-26s %8.1f MB  %s' % (n, str(list(shape)), nbytes / 1e6, who), file=sys.stderr)
    atexit.register(_dump)


def attach(t, amax):
    """Record a bound that a producing kernel left in `amax` ([1] fp32 device tensor) for tensor t."""
    try:
        setattr(t, _ATTR, (t._version, t.data_ptr(), amax, _epoch))
    except (AttributeError, RuntimeError):
        pass
    return t


def invalidate(t):
    """t was written through its raw pointer: a cached bound no longer holds."""
    if getattr(t, _ATTR, None) is not None:
        try:
            delattr(t, _ATTR)
        except AttributeError:
            pass


def tracking():
    """True while the convolution family multiplies block-scaled fp16 splits (terms = 4): producers then leave their output's bound behind."""
    import sys
    cg = sys.modules.get(__package__ + '.conv2d_gradfix')
    return cg is not None and (cg.native_conv_terms == 4 or cg.native_wrw_terms == 4)


def launch_tracking(out, call):
    """Run `call()` -- exactly ONE sgv_* launch that writes the fp32 tensor `out` -- with the library's one-shot bound side output armed
    (sgv_amax_sink): a kernel that supports it leaves max |out| behind for free and the bound is attached to `out`; any other kernel leaves the
    tensor without one (the consumer then runs `bound`, one streaming pass).  Returns what `call` returns."""
    if not (out.is_cuda and out.dtype == torch.float32 and tracking()):
        return call()
    lib = custom_ops.get_native()
    buf = zero_sink(out.device)
    lib.sgv_amax_sink(buf.data_ptr())
    try:
        rc = call()
    finally:
        taken = lib.sgv_amax_sink_consumed()
        lib.sgv_amax_sink(None)
    if taken:
        attach(out, buf[0:1])
    return rc


# ---------------------------------------------------------------------------------------------------------------------------------------
# Bounds that follow from another bound.  |FIR(x)| <= gain * sum|taps| * max|x| for every geometry of upfirdn2d (zero insertion only drops terms):
# a FIR output whose kernel has no side output (the lane-exchange kernels of the 2x up / down passes) inherits gain * sum|taps| x the bound of its
# input -- one 1-element multiply instead of a pass over the tensor.  For the normalised [1,3,3,1] low-pass that factor is 1 (down) or 4 (up, gain 4:
# loose by the zero insertion, far inside the ~2^10 a bound may be loose by, csrc/sgv_split.h).

# sum |taps| of a filter is read back ONCE per filter TENSOR OBJECT and kept on it (never while a hipGraph is being captured).  Not in a table keyed by the
# address: a temporary filter's block is handed out again by the caching allocator, and a different filter at the same address with the same version
# would inherit the old sum -- too small a bound overflows the fp16 split.


def cached(t):
    """The bound a producer or an earlier request left on t, or None."""
    c = getattr(t, _ATTR, None)
    return c[2] if (c is not None and c[0] == t._version and c[1] == t.data_ptr() and c[3] == _epoch) else None


def set_tap_sum(f, value):
    """Record sum |taps| of filter tensor f without reading it back (a caller that knows it)."""
    try:
        setattr(f, _TAPS_ATTR, (f._version, f.data_ptr(), float(value)))
    except (AttributeError, RuntimeError):
        pass


def can_inherit_through_fir(x, f2d):
    """True if `inherit_through_fir` will succeed for an output of x filtered with f2d: x carries a bound and the filter's tap sum is known (or may be read back now).
    The caller then launches WITHOUT the side output: no atomics in the streaming kernel and no fold launch behind it (round 6: 179 fold launches per iteration)."""
    if not (x.is_cuda and x.dtype == torch.float32 and tracking()) or cached(x) is None:
        return False
    rec = getattr(f2d, _TAPS_ATTR, None)
    return (rec is not None and rec[0] == f2d._version and rec[1] == f2d.data_ptr()) or not torch.cuda.is_current_stream_capturing()


def inherit_through_fir(y, x, f2d, gain):
    """Give y = upfirdn2d(x, f2d, gain=gain, ...) the bound gain * sum|f2d| * bound(x) if x has one and y has none yet.  `f2d`: the filter tensor OBJECT the
    tap sum is remembered on -- the caller's own tensor (a module buffer), not a per-call view of it: a fresh object has no record, and every miss is a
    read-back (a host / device synchronisation)."""
    if not (y.is_cuda and y.dtype == torch.float32 and tracking()) or cached(y) is not None:
        return
    bx = cached(x)
    if bx is None:
        return
    rec = getattr(f2d, _TAPS_ATTR, None)
    if rec is not None and rec[0] == f2d._version and rec[1] == f2d.data_ptr():
        tap_sum = rec[2]
    else:
        if torch.cuda.is_current_stream_capturing():
            return
        tap_sum = float(f2d.abs().sum())
        try:
            setattr(f2d, _TAPS_ATTR, (f2d._version, f2d.data_ptr(), tap_sum))
        except (AttributeError, RuntimeError):
            pass
    factor = abs(float(gain)) * tap_sum
    # factor <= 1 (the normalised low-pass at gain 1: every down-sampling / pre-convolution pass): the input's bound IS the output's -- shared, not multiplied: no launch
    # at all (the split's scale leaves a factor 2 of headroom above the bound, csrc/sgv_split.h; the FIR's own roundings are 1e-7 of it)
    attach(y, bx if factor <= 1.0001 else bx * (factor * 1.0000005))


def same_values(new, src):
    """`new` holds the values of `src` in another arrangement (a flip, a transposition, a contiguous or fp32 copy of a weight): max |new| = max |src|.  The bound is taken on
    `src` -- a module parameter keeps it until the optimiser rewrites it -- instead of once per temporary (89 separate passes over weights per training iteration,
    profiles/r06_c17_captured_census.txt)."""
    if new is src or not (tracking() and src.is_cuda and src.dtype in _DT and new.is_cuda and new.dtype in _DT) or cached(new) is not None:
        return new
    if new.dtype != src.dtype and new.dtype != torch.float32:
        return new        # a narrowing copy rounds: not the same values
    return attach(new, bound(src))


def share(alias, t):
    """`alias` is another tensor object over t's memory (an autograd node handing its input on): it carries t's bound."""
    b = cached(t)
    if b is not None and alias.data_ptr() == t.data_ptr():
        attach(alias, b)
    return alias
