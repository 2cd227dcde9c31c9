# round 6, call 3: why the product tile kernel loses 10 % when it follows itself back to back (x4 brackets) and V6 does not: per-dispatch begin / end from the kernel tracer
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c3
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && UFD_ONLY="V7 product" UFD_ONLY2="NTload0 xcd1" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- $GRAFT_REPO_ROOT/tools/ufd_lab6 96 257 1 > $GRAFT_REPO_ROOT/$OUT/lab.log 2>&1 )
find $OUT/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/trace_n96.csv
rm -rf $OUT/trace
cat $OUT/lab.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r06_c3/trace_n96.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prev_end = None
for r in rows:
    n = r['Kernel_Name']
    if 'tile' not in n: prev_end = int(r['End_Timestamp']); continue
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(('V7' if 'upfirdn2d_tile_kernel' in n else 'V6'), 'dur %.1f us' % ((en - st) / 1e3), 'gap %.1f us' % (((st - prev_end) / 1e3) if prev_end else -1))
    prev_end = en
PY
