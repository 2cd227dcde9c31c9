# round 3, GPU call 35: the round's record on the current tree:
# whole -m gpu suite, smoke(), the driver's bench command, rocprofv3 kernel stats of the step, PMC FETCH / WRITE / MFMA passes, 8 videos/GPU and g1024 lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
C=${SGV_COMMIT:-625c34c}
timeout 420 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r03_final_tests.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_final_tests.log | grep -E "passed|failed|FAILED|rror" | cut -c1-260 | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r03_smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r03_bench_final.err | tail -1 > gpurun_out/r03_bench_final.json; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03_bench_final.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'value_no_prof', 'value_strict_fp32', 'value_aug_ada', 'value_bf16_products', 'value_lowp_bf16', 'value_pl_f1')})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel', 'achieved', 'peak', 'frac', 'traffic')})
print('ufd', {k: d['roofline_upfirdn2d'].get(k) for k in ('achieved', 'frac', 'traffic')})
print('cpu', d['cpu_baseline'])
PY
B="python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof22 -- $B --steps 16 --warmup 2 > /tmp/prof22.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof22 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_bench_step_kernel_stats_final.csv
grep '"metric"' /tmp/prof22.log > gpurun_out/r03_bench_under_rocprof.json; cut -c1-160 gpurun_out/r03_bench_under_rocprof.json
head -30 gpurun_out/r03_bench_step_kernel_stats_final.csv | cut -c1-170
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc22/pmc_bench_$c -- $B --steps 2 --warmup 1 --no-prof > /tmp/pmc22_$c.log 2>&1; echo "pmc $c rc=$?"
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pmc22/pmc_bench_MFMA -- $B --steps 2 --warmup 1 --no-prof > /tmp/pmc22_MFMA.log 2>&1; echo "pmc MFMA rc=$?"
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py /tmp/pmc22 gpurun_out/r03_pmc_bench_step_FETCH_WRITE.json $C
python tools/pmc_kernel_table.py /tmp/pmc22/pmc_bench_MFMA > gpurun_out/r03_pmc_bench_step_MFMA_table.txt
grep -E "conv3x3_ws|s2_pairs|convT3x3_s2_ws|wrw3x3" gpurun_out/r03_pmc_bench_step_MFMA_table.txt | cut -c1-220

timeout 150 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --clean-steps 0 --batch-gpu 8 2>/dev/null | tail -1 > gpurun_out/r03_bench_batch8.json; cut -c1-140 gpurun_out/r03_bench_batch8.json
timeout 150 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --clean-steps 0 --batch-gpu 8 --graphs 2>/dev/null | tail -1 > gpurun_out/r03_bench_batch8_hipgraph.json; cut -c1-140 gpurun_out/r03_bench_batch8_hipgraph.json
timeout 200 python bench.py --workload g1024 --cpu-seconds 10 2>/dev/null | tail -1 > gpurun_out/r03_bench_g1024.json; cut -c1-200 gpurun_out/r03_bench_g1024.json
