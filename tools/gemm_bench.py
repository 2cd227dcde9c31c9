#!/usr/bin/env python3
"""Per-shape table of the dense 1x1 / skip GEMMs of the FFS-256 step (DiscriminatorBlock.skip, networks.py:452, via conv2d_resample.py:40-54) and of
the unfolded trajectory convolutions (motion.py), N = 96 frames: forward, data gradient and weight gradient through sgv_gemm_f32, once with the
bf16x3 member (default) and once on the exact-fp32 matrix pipe (SGV_GEMM_TERMS=0).  TFLOP/s are algorithmic (2 M N K), GB/s count every operand once.

    python tools/gemm_bench.py            # both modes (re-executes itself for the second one), prints one JSON document
"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_mode():
    import torch
    from stylegan_v_amd.torch_utils import custom_ops
    from stylegan_v_amd.torch_utils.ops import gemm
    dev = torch.device('cuda')
    n = int(os.environ.get('N', 96))
    rows = []

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        custom_ops.prof_enable(4096)
        for _ in range(reps):
            fn()
        custom_ops.prof_disable()
        prof = custom_ops.prof_collect()['gemm']
        return prof['ms'] / reps, prof['launches'] / reps

    for cin, cout, r in ((64, 128, 128), (128, 256, 64), (256, 512, 32), (512, 512, 16)):
        nn = n if r > 16 else n // 3      # below the frame-concatenation resolution the batch is videos
        x = torch.randn([nn, cin, r, r], device=dev)
        w = torch.randn([cout, cin, 1, 1], device=dev) / cin ** 0.5
        dy = torch.randn([nn, cout, r, r], device=dev)
        wt = w.reshape(cout, cin).t().reshape(cin, cout, 1, 1).contiguous()
        flops = 2.0 * nn * r * r * cin * cout
        bytes_fd = 4.0 * nn * r * r * (cin + cout)
        with torch.no_grad():
            for name, fn in (('forward', lambda: gemm.conv1x1(x, w)), ('data gradient', lambda: gemm.conv1x1(dy, wt)), ('weight gradient', lambda: gemm.conv1x1_weight_grad(dy, x))):
                ms, launches = timeit(fn)
                rows.append(dict(layer=f'skip {cin}->{cout} @ {r}^2 x {nn}', op=name, ms=ms, TFLOPs=flops / ms / 1e9, GBps=bytes_fd / ms / 1e6, launches=launches))
    a = torch.randn([32 * 66, 5632], device=dev)
    b = torch.randn([512, 5632], device=dev)
    with torch.no_grad():
        ms, launches = timeit(lambda: gemm.matmul_nt(a, b))
    rows.append(dict(layer='trajectory conv1d k=11 as [2112, 5632] x [5632, 512]', op='forward', ms=ms, TFLOPs=2.0 * 2112 * 5632 * 512 / ms / 1e9,
                     GBps=4.0 * (2112 * 5632 + 512 * 5632 + 2112 * 512) / ms / 1e6, launches=launches))
    print(json.dumps(dict(mode='fp32 MFMA (SGV_GEMM_TERMS=0)' if os.environ.get('SGV_GEMM_TERMS') == '0' else 'bf16x3', variants=custom_ops.kernel_variant_counts(), rows=rows)))


if __name__ == '__main__':
    if os.environ.get('SGV_GEMM_BENCH_CHILD'):
        run_mode()
    else:
        out = []
        for terms in ('3', '0'):
            env = dict(os.environ, SGV_GEMM_BENCH_CHILD='1', SGV_GEMM_TERMS=terms)
            res = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in res.stdout.splitlines() if l.startswith('{')]
            if not line:
                sys.exit(res.stderr[-3000:])
            out.append(json.loads(line[-1]))
        for a, b in zip(out[0]['rows'], out[1]['rows']):
            print('%-55s %-16s bf16x3 %7.3f ms %6.1f TF/s %7.1f GB/s | fp32 MFMA %7.3f ms %6.1f TF/s | x%.2f' %
                  (a['layer'], a['op'], a['ms'], a['TFLOPs'], a['GBps'], b['ms'], b['TFLOPs'], b['ms'] / a['ms']))
        print(json.dumps(out))
