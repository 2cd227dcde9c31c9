# per-dispatch durations of the stride-1 kernel variants over a few bench steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --no-prof > /tmp/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/trace_r02 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
out = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'conv3x3_ws_kernel' in n:
        key = n[n.index('conv3x3_ws_kernel'):][:40]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        out[key].append((d, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('LDS_Block_Size', '')))
for k, v in out.items():
    print(k, len(v))
    print('   ', ' '.join(f'{d:.0f}' for d, g, l in v[-40:]))
PY
