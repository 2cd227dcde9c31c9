#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 `--pmc ... --kernel-trace --output-format csv` runs.

    python tools/pmc_kernel_table.py <dir> [<dir> ...] > table.txt

Each <dir> is the -d directory of one rocprofv3 pass (its *counter_collection.csv is read)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    for ns in ('sgv_conv::', 'sgv_wrw::', 'sgv_gemm::', '(anonymous namespace)::'):
        if ns in name:
            name = name.split(ns, 1)[1]
            break
    return name.split('(')[0][:70]


def main():
    table = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for root in sys.argv[1:]:
        for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(path)):
                e = table[short(r['Kernel_Name'])][r['Counter_Name']]
                e[0] += 1
                e[1] += float(r['Counter_Value'])
    counters = sorted({c for k in table.values() for c in k})
    print('%-72s %6s ' % ('kernel', 'calls') + ' '.join('%22s' % c for c in counters))
    for k in sorted(table):
        calls = max(v[0] for v in table[k].values())
        print('%-72s %6d ' % (k, calls) + ' '.join('%22.4g' % (table[k][c][1] / table[k][c][0]) if c in table[k] else '%22s' % '-' for c in counters))


if __name__ == '__main__':
    main()
