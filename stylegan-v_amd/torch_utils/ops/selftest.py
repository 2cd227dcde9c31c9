"""Load-time self-test of the kernels whose global loads are inline assembly with hand-counted ``s_waitcnt vmcnt(N)``.

The producer / consumer members of the 3x3 family (csrc/conv3x3_ws_kernel.h, conv3x3s2_ws_kernel.h, wrw_ws_kernel.h, wrw_s2_ws_kernel.h) keep their
operand loads out of the compiler's sight so that nothing it inserts drains a pipelined chunk; the price is that their correctness rests on counts
written by hand for one compiler and one ISA.  ``pytest -m gpu`` checks those kernels
exhaustively, but a deployment does not run pytest.  So the first native convolution of a process on a device runs this once: every asm-load member
(stride 1 forward and data-gradient form, strided, transposed, both weight gradients), on small-integer data -- where the split products and the fp32
sums are exact, so the result must EQUAL torch's own convolution computed on the host, bit for bit -- and at shapes with several K chunks and several
tiles per workgroup, so that the software pipelines actually wrap around.  Eight launches and a few seconds of host time (the references: ~90 GFLOP on
the host cores, shared out between the ranks of a node), once.

A mismatch means the hand-counted waits do not hold on this stack: the product does not continue on kernels that returned a wrong number.
``SGV_SELFTEST=fallback`` instead switches the 3x3 family to the vendor library for the process (terms = 0), loudly; ``SGV_SELFTEST=0`` skips the
test (benchmarks that time the first call).  Nothing here imports ``oracle/``: the reference is torch's own convolution on the CPU.
"""

import os
import sys

import torch

_state = {}      # device index -> 'ok' | 'fallback' | 'running'


N, CIN, COUT, RES = 72, 64, 128, 32      # 4 K chunks; 288 ... 576 tiles on <= 256 persistent workgroups: the pipelines wrap around chunks AND tiles


def _cases():
    g = torch.Generator().manual_seed(20260926)

    def ints(shape, lim):
        return torch.randint(-lim, lim + 1, shape, generator=g).float()

    x = ints([N, CIN, RES, RES], 3)
    yield 'stride 1', (x, ints([COUT, CIN, 3, 3], 2)), (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
    yield 'stride 1, data-gradient form', (x, ints([CIN, COUT, 3, 3], 2)), (True, (1, 1), (1, 1), (0, 0), (1, 1), 1)
    yield 'stride 2', (ints([N, CIN, 2 * RES + 1, 2 * RES + 1], 3), ints([COUT, CIN, 3, 3], 2)), (False, (2, 2), (0, 0), (0, 0), (1, 1), 1)
    yield 'transposed stride 2', (x, ints([CIN, CIN, 3, 3], 2)), (True, (2, 2), (0, 0), (0, 0), (1, 1), 1)


def _reference(x, w, cfg):
    """torch's CPU convolution in fp32: on these integers every product and every partial sum is an integer below 2^24, so fp32 is exact in any order."""
    transposed, stride, padding = cfg[0], cfg[1], cfg[2]
    op = torch.nn.functional.conv_transpose2d if transposed else torch.nn.functional.conv2d
    return op(x, w, stride=stride, padding=padding)


def _run_cases(device, cg, bad):
    with torch.no_grad():
        for name, (x, w), cfg in _cases():
            xd, wd = x.to(device), w.to(device)
            if cg._native_conv_ok(xd, wd, cfg):
                got = cg._native_conv(xd, wd, cfg).cpu()
                if not torch.equal(got, _reference(x, w, cfg)):
                    bad.append(f'{name} (terms {cg.native_conv_terms})')
            # the weight gradient of the same layer: dy = an integer tensor of the output's shape
            ref_w = torch.zeros(w.shape, requires_grad=True)
            with torch.enable_grad():
                y = _reference(x, ref_w, cfg)
            dy = torch.randint(-2, 3, y.shape, generator=torch.Generator().manual_seed(7)).float()
            (want,) = torch.autograd.grad(y, ref_w, dy)
            dyd = dy.to(device)
            if cg._native_wrw_ok(dyd, xd, cfg, tuple(w.shape)):
                got = cg._native_wrw(dyd, xd, cfg, tuple(w.shape)).cpu()
                if not torch.equal(got, want):
                    bad.append(f'weight gradient, {name} (terms {cg.native_wrw_terms})')


def run(device):
    """Run once per process and device; returns 'ok', 'fallback' (vendor library from here on), 'off' (SGV_SELFTEST=0), 'running' (re-entered by the
    test's own launches) or 'deferred' (a hipGraph is being captured: next call).  Raises RuntimeError on a mismatch unless SGV_SELFTEST=fallback."""
    key = torch.device(device).index or 0
    if key in _state:
        return _state[key]
    mode = os.environ.get('SGV_SELFTEST', '1')
    if mode == '0':
        _state[key] = 'off'
        return 'off'
    if torch.cuda.is_current_stream_capturing():
        return 'deferred'
    _state[key] = 'running'          # (the test's own convolutions re-enter the dispatch)
    from . import conv2d_gradfix as cg
    bad = []
    # one process per GPU: every rank of a node runs this at the same moment, and the CPU references (~90 GFLOP per process) of 8 ranks on all cores each would
    # oversubscribe the host 8 x -- each rank takes its share of the cores for the duration
    local_world = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1))
    threads = torch.get_num_threads()
    if local_world > 1:
        torch.set_num_threads(max(1, threads // local_world))
    try:
        _run_cases(device, cg, bad)
    except BaseException:
        # anything but a verdict (out of memory, a launch error, an interrupt) must not leave 'running' behind: every later call would return it,
        # and the process would go on with unverified kernels and no warning.  The next convolution runs the test again.
        _state.pop(key, None)
        print('[sgv] libsgv_hip self-test did not complete; it will run again at the next convolution', file=sys.stderr, flush=True)
        raise
    finally:
        if local_world > 1:
            torch.set_num_threads(threads)
    if not bad:
        _state[key] = 'ok'
        return 'ok'
    msg = ('libsgv_hip self-test: the producer / consumer convolution kernels (inline-asm loads, hand-counted waits) returned wrong results on exact integer '
           'data: ' + '; '.join(bad) + '.  This compiler / driver stack is not one the kernels were validated on (custom_ops.VALIDATED_COMPILERS).')
    if mode == 'fallback':
        print('[sgv] ' + msg + '  SGV_SELFTEST=fallback: the 3x3 family runs on the vendor library in this process.', file=sys.stderr, flush=True)
        cg.native_conv_terms = cg.native_wrw_terms = 0
        _state[key] = 'fallback'
        return 'fallback'
    del _state[key]
    raise RuntimeError(msg + '  Set SGV_SELFTEST=fallback to continue on the vendor library, or SGV_CONV_TERMS=0 SGV_WRW_TERMS=0.')
