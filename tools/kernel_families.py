#!/usr/bin/env python3
"""Per-family table of a `rocprofv3 --kernel-trace --stats` CSV of bench.py (profiles/r0N_bench_step_kernel_stats_final.csv): ms per step, share, launches
per step; own kernels against torch element-wise / vendor library.

    python tools/kernel_families.py profiles/r03_bench_step_kernel_stats_final.csv [steps]

`steps` = iterations inside the profiled command (timed + warm-up; default 18 = `--steps 16 --warmup 2`)."""
import collections
import csv
import sys

FAMILIES = (   # first match wins
    ('3x3 stride 1 (forward / data gradient / fused layer)', ('conv3x3_ws_kernel', 'conv3x3_kernel<')),
    ('3x3 weight gradient, stride 1', ('wrw3x3_ws_kernel', 'wrw3x3_kernel')),
    ('3x3 weight gradient, stride 2', ('wrw3x3_s2',)),
    ('3x3 transposed stride 2 (+ edge strips)', ('convT3x3',)),
    ('3x3 strided', ('conv3x3_s2',)),
    ('3x3 on 16^2 / 8^2 images', ('conv3x3_small',)),
    ('weight preparation', ('prep_weights',)),
    ('upfirdn2d', ('upfirdn2d',)),
    ('dense 1x1 / trajectory GEMM', ('sgv_gemm',)),
    ('dense layers (fc)', ('fc_kernel',)),
    ('ToRGB / fromRGB streams', ('pw_',)),
    ('bias_act', ('bias_act_kernel',)),
    ('modulation / fused-layer element-wise backward / multi-tensor', ('act_grad_scale', 'scale_dot', 'scale_channels', 'plane_dot', 'demod_coefs', 'weight_sqsum', 'multi_')),
    ('temporal encoder / resampling', ('time_encode', 'affine_resample')),
    ('torch element-wise, reductions, fills, optimiser', ('at::native', 'rocclr')),
    ('vendor library (MIOpen / rocBLAS / hipBLASLt)', ('miopen', 'igemm', 'batched_transpose', 'Cijk', 'SubTensor', 'gemv', 'rocblas')),
)


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return 'other'


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 18.0
    ms, calls = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        f = family(r['Name'])
        ms[f] += int(r['TotalDurationNs']) / 1e6 / steps
        calls[f] += int(r['Calls']) / steps
    total = sum(ms.values())
    print('%-66s %9s %7s %10s' % ('family', 'ms/step', 'share', 'launches'))
    for f, v in ms.most_common():
        print('%-66s %9.2f %6.1f%% %10.1f' % (f, v, 100 * v / total, calls[f]))
    print('%-66s %9.2f %6.1f%% %10.1f' % ('all kernels', total, 100.0, sum(calls.values())))


if __name__ == '__main__':
    main()
