#!/usr/bin/env python3
"""Time the one-kernel ADA adjoint (and forward) per SAMPLE over many maps drawn the way the bgc pipeline draws them at p = 1: which maps are slow?

    python tools/ada_adjoint_sweep.py [--maps 256] [--static 0]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ada_bench import drawn_maps  # noqa: E402
from stylegan_v_amd.training.augment import AugmentPipe, BGC  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--maps', type=int, default=256)
    ap.add_argument('--static', type=int, default=0)
    args = ap.parse_args()
    dev = torch.device('cuda')
    gen = torch.Generator().manual_seed(1)
    pipe = AugmentPipe(**BGC).to(dev)
    pipe.static_margin = bool(args.static)
    x = torch.randn([1, 9, 256, 256], generator=gen).to(dev)
    v = torch.randn(x.shape, generator=gen).to(dev)
    maps = drawn_maps(args.maps, 256, gen)
    rows = []
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for i in range(args.maps):
        g_inv = maps[i:i + 1].to(dev) if args.static else maps[i:i + 1]
        xg = x.clone().requires_grad_(True)
        for rep in range(2):
            e0.record()
            y = pipe._resample(xg, g_inv)
            e1.record()
            torch.autograd.grad(y, xg, v)
            e2.record()
            e2.synchronize()
        m = maps[i]
        det = float(m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0])
        rows.append((e1.elapsed_time(e2) * 1e3, e0.elapsed_time(e1) * 1e3, i, det, [round(float(t), 3) for t in m[:2].flatten()]))
    rows.sort(reverse=True)
    bw = sorted(r[0] for r in rows)
    fw = sorted(r[1] for r in rows)
    print(f'# {args.maps} maps, one sample x 9 channels x 256^2 per call, static margin {args.static}; backward us: median {bw[len(bw)//2]:.0f}, p90 {bw[int(len(bw)*0.9)]:.0f}, max {bw[-1]:.0f}; '
          f'forward us: median {fw[len(fw)//2]:.0f}, p90 {fw[int(len(fw)*0.9)]:.0f}, max {fw[-1]:.0f}')
    for r in rows[:12]:
        print(f'backward {r[0]:9.0f} us  forward {r[1]:9.0f} us  map {r[2]:4d}  det {r[3]:7.3f}  G_inv rows {r[4]}')


if __name__ == '__main__':
    main()
