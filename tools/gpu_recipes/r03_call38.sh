# round 3, GPU call 38: the round's record on the final tree: whole -m gpu suite, smoke(), the driver's bench command, rocprofv3 kernel stats of the step,
# the 1024^2 synthesis line with its kernel stats and PMC FETCH / WRITE passes (the step's PMC passes of call 35 stay: its kernels did not change)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
C=${SGV_COMMIT:-69da6b6}
timeout 400 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r03_final_tests.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_final_tests.log | grep -E "passed|failed|FAILED|rror" | cut -c1-260 | tail -8
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r03_smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r03_bench_final.err | tail -1 > gpurun_out/r03_bench_final.json; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03_bench_final.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'value_no_prof', 'value_strict_fp32', 'value_aug_ada', 'value_bf16_products', 'value_lowp_bf16', 'value_pl_f1')})
print('roofline', {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'traffic')}, 'ufd', {k: d['roofline_upfirdn2d'].get(k) for k in ('achieved', 'frac', 'traffic')})
PY
B="python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0"
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof38 -- $B --steps 16 --warmup 2 > /tmp/prof38.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof38 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_bench_step_kernel_stats_final.csv
grep '"metric"' /tmp/prof38.log > gpurun_out/r03_bench_under_rocprof.json; cut -c1-120 gpurun_out/r03_bench_under_rocprof.json
G="python $GRAFT_REPO_ROOT/bench.py --workload g1024 --cpu-seconds 0"
cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof38g -- $G --steps 16 --warmup 3 > /tmp/prof38g.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc38/pmc_bench_$c -- $G --steps 2 --warmup 1 --no-prof > /tmp/pmc38_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
find /tmp/prof38g -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_g1024_kernel_stats.csv
python tools/pmc_summary.py /tmp/pmc38 gpurun_out/r03_pmc_g1024_FETCH_WRITE.json $C > /dev/null
cp gpurun_out/r03_pmc_g1024_FETCH_WRITE.json profiles/
timeout 150 python bench.py --workload g1024 --cpu-seconds 10 2>/dev/null | tail -1 > gpurun_out/r03_bench_g1024.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_g1024.json')); print('g1024', round(d['value'],1), round(d['value_no_prof'],1), {k:d['roofline'][k] for k in ('achieved','frac','traffic')}, d['cpu_baseline']['value'])"
