"""Report for tools/captured_census.py: kernels per replayed main iteration, by name, small ones (< 12 us) first.  Usage: captured_census_report.py trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'scan' in r['Kernel_Name'].lower() or 'cumsum' in r['Kernel_Name'].lower()]
print('marker kernels at', marks[-6:], 'of', len(rows))
# the last two marker groups bracket the replays
groups = []
for i in marks:
    if groups and i - groups[-1][-1] < 8:
        groups[-1].append(i)
    else:
        groups.append([i])
lo, hi = groups[-2][-1] + 1, groups[-1][0]
seg = rows[lo:hi]
its = 4.0
t_total = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
print(f'{len(seg) / its:.0f} kernels per iteration, {t_total / its:.2f} ms per iteration, kernels busy {busy / its:.2f} ms')
small = collections.defaultdict(lambda: [0, 0.0])
big = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    (small if d < 12 else big)[r['Kernel_Name'][:120]][0] += 1
    (small if d < 12 else big)[r['Kernel_Name'][:120]][1] += d
print(f'small (< 12 us): {sum(v[0] for v in small.values()) / its:.0f} launches, {sum(v[1] for v in small.values()) / its / 1e3:.2f} ms per iteration')
for k, v in sorted(small.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f'{v[0] / its:7.1f}/it {v[1] / v[0]:6.1f} us  {k}')
print('large:')
for k, v in sorted(big.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{v[0] / its:7.1f}/it {v[1] / v[0]:8.1f} us {v[1] / its / 1e3:7.2f} ms/it  {k}')
