#!/usr/bin/env python3
"""The ADA geometric block at the training step's size (32 clips x 9 channels x 256^2): one-kernel forward / one-kernel adjoint against the four-pass
composition's forward / backward, identity maps (p = 0: what a fresh run executes) and maps drawn the way the bgc pipeline draws them at p = 1.

    python tools/ada_bench.py [--n 32] [--c 9] [--res 256] [--static 1]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_v_amd.torch_utils import custom_ops  # noqa: E402
from stylegan_v_amd.training.augment import AugmentPipe, BGC  # noqa: E402


def drawn_maps(n, res, gen):
    """G_inv of the bgc pipeline at p = 1 (augment.py:226-268): x-flip, 90-degree rotations, integer translation, isotropic scale (lognormal, std 0.2 octaves),
    rotation (uniform in +-pi), anisotropic scale, second rotation, fractional translation."""
    out = []
    for _ in range(n):
        m = torch.eye(3, dtype=torch.float64)

        def push(a):
            nonlocal m
            m = m @ torch.tensor(a, dtype=torch.float64)
        r = lambda: float(torch.rand([], generator=gen))       # noqa: E731
        nrm = lambda: float(torch.randn([], generator=gen))    # noqa: E731
        push([[1 - 2 * (r() < 0.5), 0, 0], [0, 1, 0], [0, 0, 1]])
        a = math.pi / 2 * int(r() * 4)
        push([[math.cos(a), math.sin(-a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        push([[1, 0, -round((r() * 2 - 1) * 0.125 * res)], [0, 1, -round((r() * 2 - 1) * 0.125 * res)], [0, 0, 1]])
        s = 2 ** (nrm() * 0.2)
        push([[1 / s, 0, 0], [0, 1 / s, 0], [0, 0, 1]])
        a = (r() * 2 - 1) * math.pi
        push([[math.cos(a), math.sin(-a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        s = 2 ** (nrm() * 0.2)
        push([[1 / s, 0, 0], [0, s, 0], [0, 0, 1]])
        a = (r() * 2 - 1) * math.pi
        push([[math.cos(a), math.sin(-a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        push([[1, 0, -nrm() * 0.125 * res], [0, 1, -nrm() * 0.125 * res], [0, 0, 1]])
        out.append(m.float())
    return torch.stack(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=32)
    ap.add_argument('--c', type=int, default=9)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--static', type=int, default=1)
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    dev = torch.device('cuda')
    gen = torch.Generator().manual_seed(0)
    x = torch.randn([args.n, args.c, args.res, args.res], generator=gen).to(dev)
    v = torch.randn(x.shape, generator=gen).to(dev)
    fused, comp = AugmentPipe(**BGC).to(dev), AugmentPipe(**BGC).to(dev)
    comp.fused_geometric = False
    fused.static_margin = comp.static_margin = bool(args.static)
    nbytes = 2 * x.numel() * 4
    for name, g_inv in (('identity maps', torch.eye(3).repeat(args.n, 1, 1)), ('maps drawn at p = 1', drawn_maps(args.n, args.res, gen))):
        g_inv = g_inv.to(dev) if args.static else g_inv
        cases = []
        for label, pipe in (('one kernel', fused), ('composition', comp)):
            xg = x.clone().requires_grad_(True)
            y = pipe._resample(xg, g_inv)
            cases.append((label + ' forward', lambda pipe=pipe, xg=xg: pipe._resample(xg, g_inv)))
            cases.append((label + ' backward', lambda y=y, xg=xg: torch.autograd.grad(y, xg, v, retain_graph=True)))
        ms = {c[0]: [] for c in cases}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            for _, fn in cases:
                fn()
        torch.cuda.synchronize()
        for rd in range(args.rounds):
            for label, fn in cases:
                for _ in range(4):
                    fn()
                before = custom_ops.launch_count()
                e0.record()
                for _ in range(8):
                    fn()
                e1.record()
                e1.synchronize()
                ms[label].append((e0.elapsed_time(e1) / 8, (custom_ops.launch_count() - before) // 8))
        print(f'# {name}; {args.n} x {args.c} x {args.res}^2, static margin {args.static}; algorithmic bytes per pass {nbytes/1e6:.1f} MB')
        for label, _ in cases:
            t = sorted(ms[label])
            med = t[len(t) // 2]
            print(f'{label:28s} {med[0]*1e3:9.1f} us  ({med[1]} launches)  {nbytes/med[0]/1e6:8.1f} GB/s algorithmic')


if __name__ == '__main__':
    main()
