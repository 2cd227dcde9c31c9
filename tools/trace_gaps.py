#!/usr/bin/env python3
"""Idle time on the device between consecutive kernels of a rocprofv3 --kernel-trace CSV: total, and attributed to the kernel that FOLLOWS each gap.

    python tools/trace_gaps.py <kernel_trace.csv> [min_gap_us]"""
import collections
import csv
import sys


def main():
    path, min_gap = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path))]
    rows.sort()
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps, by = 0, collections.defaultdict(lambda: [0, 0.0])
    dur = collections.defaultdict(lambda: [0, 0.0])
    last_end = rows[0][1]
    for s, e, k in rows:
        name = k.split('(')[0][-70:]
        dur[name][0] += 1
        dur[name][1] += (e - s) / 1e3
        g = (s - last_end) / 1e3
        if g > min_gap:
            gaps += g
            by[name][0] += 1
            by[name][1] += g
        last_end = max(last_end, e)
    print('kernels %d  span %.1f ms  busy %.1f ms  idle in gaps > %.0f us: %.1f ms' % (len(rows), span / 1e6, busy / 1e6, min_gap, gaps / 1e3))
    print('-- idle attributed to the kernel after the gap')
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print('%9.2f ms %6d gaps  %s' % (t / 1e3, n, k))
    print('-- tiny kernels (< 10 us average)')
    tiny = [(k, v) for k, v in dur.items() if v[1] / v[0] < 10]
    print('%d launches, %.2f ms busy' % (sum(v[0] for _, v in tiny), sum(v[1] for _, v in tiny) / 1e3))


if __name__ == '__main__':
    main()
