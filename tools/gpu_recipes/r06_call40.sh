#!/bin/bash
# call 40: kernel statistics of the captured step with and without aug=ada: which kernels does the augmentation add?
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c40
export TMPDIR=/tmp
for aug in noaug ada; do
  ( cd /tmp && rm -rf /tmp/prof_$aug && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$aug -- python $GRAFT_REPO_ROOT/tools/ada_step_bench.py --aug $aug --graphs 1 --steps 6 > /tmp/prof_$aug.log 2>&1 )
  tail -1 /tmp/prof_$aug.log
  f=$(find /tmp/prof_$aug -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/c40/kernel_stats_$aug.csv
done
python - <<'PY'
import csv
def load(p):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(p))}
a, b = load('gpurun_out/c40/kernel_stats_noaug.csv'), load('gpurun_out/c40/kernel_stats_ada.csv')
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, cb - ca, k))
rows.sort(reverse=True)
print('total difference (ms over the whole run):', sum(r[0] for r in rows) / 1e6)
for d, c, k in rows[:28]:
    print(f'{d/1e6:9.2f} ms  {c:6d} calls  {k[:120]}')
PY
