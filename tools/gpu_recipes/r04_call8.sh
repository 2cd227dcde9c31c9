# round 4, GPU call 8: where the aug=ada step spends its extra time (rocprofv3 kernel stats of the ada step next to the plain one), and the upfirdn2d
# micro-benchmark through the op API / C ABI at N = 32 (the 1.078 GB headline call) and N = 96
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 120 python tools/ops_bench.py --only upfirdn2d --frames 32 --reps 20 --json gpurun_out/r04_ops_bench_upfirdn2d_n32.json 2>/dev/null | tail -26
timeout 120 python tools/ops_bench.py --only upfirdn2d --frames 96 --reps 12 --json gpurun_out/r04_ops_bench_upfirdn2d_n96.json 2>/dev/null | head -6
B="python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --clean-steps 0 --no-prof"
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8a -- $B --steps 8 --warmup 2 --aug ada > /tmp/prof8a.log 2>&1; echo "rocprof ada rc=$?"
cd $GRAFT_REPO_ROOT
find /tmp/prof8a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_c8_ada_step_kernel_stats.csv
grep '"metric"' /tmp/prof8a.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ada step under rocprof', round(d['value'],1), round(d['ms_per_step'],1))"
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r04_c8_ada_step_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot / 1e6 / 10)
for r in rows[:45]:
    print('%7.2f ms/step %6d calls %9.1f us  %s' % (float(r['TotalDurationNs']) / 1e6 / 10, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
