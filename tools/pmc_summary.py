#!/usr/bin/env python3
"""Summarise rocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` counter-collection CSVs into the per-kernel JSON bench.py reads.

    python tools/pmc_summary.py <dir with pmc_bench_FETCH_SIZE/ and pmc_bench_WRITE_SIZE/> <out.json> [commit]

Collected in separate passes (TCC slot budget, MI355X_MICROARCH.md "rocprofv3 PMC slots") by tools/gpu_recipes/pmc_fetch_write_passes.sh."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root, out_path = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else os.environ.get('SGV_COMMIT', 'unknown')
    # `csrc_digest`: md5 over the kernel sources the counters were collected on (custom_ops.source_digest) -- bench.py refuses a file whose digest is not
    # the one of the library it is timing (there is no .git on the GPU box to compare commits with)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stylegan_v_amd.torch_utils import custom_ops
    out = {'commit': commit, 'csrc_digest': custom_ops.source_digest()}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        fs = glob.glob(os.path.join(root, f'pmc_bench_{c}', '*', '*counter_collection.csv'))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(fs[0])):
            k = r['Kernel_Name']
            key = None
            for ns in ('(anonymous namespace)::', 'sgv_conv::', 'sgv_wrw::', 'sgv_gemm::', 'sgv_fck::'):
                if ns in k:
                    key = k.split(ns)[1].split('(')[0][:80]
                    break
            if key is None:
                continue
            agg[key][0] += 1
            agg[key][1] += float(r['Counter_Value'])
        out[c] = {k: dict(launches=v[0], total_KB=v[1]) for k, v in agg.items()}
    with open(out_path, 'w') as fh:
        json.dump(out, fh, indent=1)
    print({k: v for k, v in out['FETCH_SIZE'].items() if 'conv' in k or 'wrw' in k or 'upfirdn' in k})


if __name__ == '__main__':
    main()
