#!/usr/bin/env python3
"""The members of the upfirdn2d family at the sizes the training step runs them (96 / 192 frames), raw C-ABI launches, settled device.

    python tools/fir_bench.py [--frames 96] [--rounds 5] [--only substring]

One row per (call, width): plain FIR passes, the fused modes 1-4 (sgv_upfirdn2d_fused) and the 2x geometries.  Protocol as tools/ufd_lab6.hip: a case gets WARM
untimed launches and REPS launches inside one event bracket per round; the rounds cycle through all cases (A B C A B C ...), the table prints the median round
(the first ~30 launches after an idle period run slow on this part: profiles/r06_c3_transient_per_dispatch.txt).  A/B across kernel forms: run the script twice,
with and without SGV_UFD_TILE2X=0 / SGV_UFD_TILE_EPI2=0 / SGV_TILE_XCD=0, inside ONE gpurun call.
Algorithmic bytes: every tensor the call must read or write, once."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_v_amd.torch_utils import custom_ops  # noqa: E402
from stylegan_v_amd.torch_utils.ops import upfirdn2d  # noqa: E402
from stylegan_v_amd.torch_utils.ops.fused_fir_act import _ufd_params  # noqa: E402

WARM, REPS = 24, 24


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=96)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--only', type=str, default=None)
    ap.add_argument('--widths', type=str, default='256,128,64')
    ap.add_argument('--amax', type=int, default=0, help='arm the bound side output (sgv_amax_sink) in front of every launch, as the training step does for tensors that feed a split-fp16 product')
    ap.add_argument('--interleave', type=int, default=0, help='matrix-bound launches (8192^3 bf16 products, ~0.6 ms each) in front of EVERY timed FIR launch: the regime of the training step, where a streaming pass follows kernels that hold the package at its power cap')
    args = ap.parse_args()
    dev = torch.device('cuda')
    lib = custom_ops.get_native()
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    stream = torch._C._cuda_getCurrentRawStream(0)
    N = args.frames
    cases = []

    sink = torch.zeros([1 + 4096], device=dev)

    def add(name, nbytes, launch, keep):
        if args.only and args.only not in name:
            return
        if args.amax:
            inner = launch

            def launch(i, inner=inner):
                lib.sgv_amax_sink(sink.data_ptr())
                inner(i)
                lib.sgv_amax_sink(None)
        cases.append(dict(name=name, bytes=nbytes, launch=launch, keep=keep, ms=[]))

    for r in [int(w) for w in args.widths.split(',')]:
        c = 64 * 256 // r
        nsets = 2 if N * c * r * r * 4 > 300e6 else 4
        big = [torch.randn([N, c, r + 1, r + 1], device=dev) for _ in range(nsets)]      # 2r+1-shaped tensors
        sml = [torch.randn([N, c, r, r], device=dev) for _ in range(nsets)]             # 2r-shaped
        half = [torch.randn([N, c, r // 2, r // 2], device=dev) for _ in range(nsets)]
        sc = torch.rand([N, c], device=dev) + 0.5
        b = torch.randn([c], device=dev)
        sums = torch.zeros([2, N * c], device=dev)
        nb, ns, nh = big[0].numel() * 4, sml[0].numel() * 4, half[0].numel() * 4

        def plain(x, y, pads, up=1, down=1, flip=False, gain=1.0):
            ps = [_ufd_params(xi, f, yi, pads, flip, gain, up=up, down=down) for xi, yi in zip(x, y)]
            return lambda i, ps=ps, n=nsets: custom_ops.check(lib.sgv_upfirdn2d(ps[i % n], 0, stream), lib)

        def fused(x, y, pads, epi_of, up=1, down=1, flip=False, gain=1.0):
            ps = [_ufd_params(xi, f, yi, pads, flip, gain, up=up, down=down) for xi, yi in zip(x, y)]
            es = [epi_of(i) for i in range(nsets)]
            return lambda i, ps=ps, es=es, n=nsets: custom_ops.check(lib.sgv_upfirdn2d_fused(ps[i % n], es[i % n], 0, stream), lib)

        keep = (big, sml, half, sc, b, sums)
        add(f'FIR {r+1}->{r} plain', nb + ns, plain(big, sml, (1, 1, 1, 1), gain=4.0), keep)
        add(f'FIR {r}->{r+1} plain', nb + ns, plain(sml, big, (2, 2, 2, 2)), keep)
        add(f'mode1 {r+1}->{r} (FIR, *scale, +bias, lrelu)', nb + ns,
            fused(big, sml, (1, 1, 1, 1), lambda i: custom_ops.FirEpilogue(1, sc.data_ptr(), b.data_ptr(), None, None, None, 3, 0.2, 2 ** 0.5, -1.0), gain=4.0), keep)
        yref_s = [torch.randn([N, c, r, r], device=dev) for _ in range(nsets)]
        add(f'mode2 {r}->{r+1} (act-grad prologue, plane sums)', 2 * ns + nb,
            fused(sml, big, (2, 2, 2, 2), lambda i: custom_ops.FirEpilogue(2, sc.data_ptr(), None, yref_s[i].data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(), 3, 0.2, 2 ** 0.5, -1.0),
                  flip=True, gain=4.0), keep + (yref_s,))
        add(f'mode3 {r+1}->{r} (act-grad epilogue)', nb + 2 * ns,
            fused(big, sml, (1, 1, 1, 1), lambda i: custom_ops.FirEpilogue(3, None, None, yref_s[i].data_ptr(), sums[0].data_ptr(), None, 3, 0.2, 2 ** 0.5, -1.0), flip=True), keep + (yref_s,))
        add(f'down2 {r}->{r//2}', ns + nh, plain(sml, half, (1, 1, 1, 1), down=2), keep)
        add(f'up2 {r//2}->{r}', ns + nh, plain(half, sml, (2, 1, 2, 1), up=2, flip=True), keep)
        add(f'mode4 up2 {r//2}->{r} + addend', 2 * ns + nh,
            fused(half, sml, (2, 1, 2, 1), lambda i: custom_ops.FirEpilogue(4, None, None, yref_s[i].data_ptr(), None, None, 1, 0.0, 1.0, -1.0), up=2, flip=True), keep + (yref_s,))

    for c in cases:
        c['launch'](0)
    torch.cuda.synchronize()
    for w in range(60):
        cases[0]['launch'](w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if args.interleave:
        ma = torch.randn([8192, 8192], device=dev, dtype=torch.bfloat16)
        mb = torch.randn([8192, 8192], device=dev, dtype=torch.bfloat16)
        mc = torch.empty([8192, 8192], device=dev, dtype=torch.bfloat16)
    for rd in range(args.rounds):
        for c in cases:
            if args.interleave:
                evs = []
                for q in range(REPS):
                    for _ in range(args.interleave):
                        torch.mm(ma, mb, out=mc)
                    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    c['launch'](q)
                    b_.record()
                    evs.append((a, b_))
                torch.cuda.synchronize()
                ts = sorted(a.elapsed_time(b_) for a, b_ in evs)
                c['ms'].append(ts[len(ts) // 2])
                continue
            for w in range(WARM):
                c['launch'](w)
            e0.record()
            for q in range(REPS):
                c['launch'](q)
            e1.record()
            e1.synchronize()
            c['ms'].append(e0.elapsed_time(e1) / REPS)
    if args.amax:
        print('# bound side output armed for every launch (the one-workgroup fold kernel behind each launch is inside the bracket)')
    switches = ' '.join(f'{k}={v}' for k, v in sorted(os.environ.items()) if k.startswith('SGV_'))
    print(f'# frames {N}; {switches or "default dispatch"}; ' + (f'every timed launch behind {args.interleave} matrix-bound launches (8192^3 bf16), one event pair per launch, median of {REPS}, median of {args.rounds} rounds' if args.interleave else f'median of {args.rounds} rounds of {REPS} launches behind {WARM} warm ones'))
    for c in cases:
        t = sorted(c['ms'])
        med = t[len(t) // 2]
        print(f"{c['name']:52s} {med*1e3:9.1f} us (min {t[0]*1e3:8.1f} max {t[-1]*1e3:8.1f})  {c['bytes']/med/1e6:8.1f} GB/s  {c['bytes']/med/1e6/80:5.1f}% of 8 TB/s")


if __name__ == '__main__':
    main()
