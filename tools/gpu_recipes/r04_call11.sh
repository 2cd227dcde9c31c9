# round 4, GPU call 11: where the aug=ada step loses its 10.7 ms: kernel trace with timestamps of the ada step and of the plain step (idle gaps, tiny launches)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --no-prof"
for aug in ada noaug; do
  rm -rf /tmp/tr_$aug
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$aug -- python $GRAFT_REPO_ROOT/bench.py $OFF --aug $aug --steps 8 --warmup 4 > $GRAFT_REPO_ROOT/gpurun_out/r04_c11_$aug.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04_c11_$aug.err )
  f=$(ls /tmp/tr_$aug/*/*kernel_trace.csv | head -1)
  echo "== $aug: $(grep -o '"value": [0-9.]*' gpurun_out/r04_c11_$aug.json | head -1)"
  python tools/trace_gaps.py $f 5 | tee gpurun_out/r04_c11_gaps_$aug.txt
  python - "$f" $aug <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d[r['Kernel_Name'].split('(')[0][-90:]][0] += 1
    d[r['Kernel_Name'].split('(')[0][-90:]][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
with open('gpurun_out/r04_c11_kernels_%s.txt' % sys.argv[2], 'w') as fh:
    for k, (n, t) in sorted(d.items(), key=lambda kv: -kv[1][1]):
        fh.write('%10.3f ms %6d  %s\n' % (t, n, k))
PY
done
