# round 6, THE record on the final kernel sources: whole -m gpu suite + smoke, counter passes (FETCH_SIZE / WRITE_SIZE, SQ matrix-pipe counters: separate --pmc runs
# with --kernel-trace only) -> digest-stamped summary, the driver's bench command with that summary in place, rocprofv3 --kernel-trace --stats of the step, the
# upfirdn2d calls through the C ABI (tools/ops_bench.py: settled / cold) and at the step's sizes (tools/fir_bench.py), the eager step, config 4's per-GPU workload
# (8 videos), the 1024^2 synthesis workload, the census of a captured main iteration
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_final
mkdir -p $OUT/tables
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; print('csrc digest', c.source_digest()); sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
OFF="--strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
# 1. tests + smoke
SGV_ERROR_TABLE_DIR=$OUT/tables timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# 2. counter passes (own runs: --pmc with --kernel-trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_bench_$c -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 2 --warmup 1 --cpu-seconds 0 --no-prof $OFF > $GRAFT_REPO_ROOT/$OUT/pmc_bench_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT $OUT/r06_pmc_bench_step_FETCH_WRITE.json "round 6 final sources" > $OUT/pmc_summary.log 2>&1; tail -2 $OUT/pmc_summary.log | cut -c1-300
cp $OUT/r06_pmc_bench_step_FETCH_WRITE.json profiles/
rm -rf $OUT/pmc_bench_FETCH_SIZE $OUT/pmc_bench_WRITE_SIZE
rm -rf /tmp/pmcsq
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pmcsq/mfma -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 2 --warmup 1 --cpu-seconds 0 --no-prof $OFF > /tmp/pmcsq_mfma.log 2>&1; echo "pmc MFMA rc=$?" )
python tools/pmc_kernel_table.py /tmp/pmcsq/mfma > $OUT/r06_pmc_bench_step_MFMA_table.txt 2>/dev/null
# 3. the driver's command
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
cp bench_detail.json $OUT/bench_driver_cmd_detail.json
grep "per-iteration" $OUT/bench_driver_cmd.err > $OUT/bench_driver_cmd_steps.log
# 4. kernel stats of the step (companions off so that the table is the headline step's)
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 $OFF > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.err )
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r06_bench_step_kernel_stats_final.csv
rm -rf $OUT/stats
# 5. the upfirdn2d family through the C ABI
timeout 300 python tools/ops_bench.py --frames 32 --only upfirdn2d --json $OUT/r06_ops_bench_upfirdn2d_n32.json > $OUT/r06_ops_bench_upfirdn2d_n32.log 2> $OUT/ops_bench.err; head -4 $OUT/r06_ops_bench_upfirdn2d_n32.log | cut -c1-200
timeout 300 python tools/fir_bench.py --frames 96 > $OUT/r06_fir_bench_n96.log 2>> $OUT/ops_bench.err
timeout 300 python tools/fir_bench.py --frames 96 --amax 1 > $OUT/r06_fir_bench_n96_bound_armed.log 2>> $OUT/ops_bench.err
# 6. other configurations
timeout 300 python bench.py --eager --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_eager.json 2> $OUT/bench_eager.err
timeout 300 python bench.py --batch-gpu 8 --eager --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_batch8_eager.json 2> $OUT/bench_batch8_eager.err
timeout 300 python bench.py --batch-gpu 8 --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_batch8_graphs.json 2> $OUT/bench_batch8_graphs.err
timeout 300 python bench.py --batch-gpu 8 --graphs --lowp bf16 --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_batch8_lowp_bf16_graphs.json 2> $OUT/bench_batch8_lowp_bf16_graphs.err
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/g1024/pmc_bench_$c -- python $GRAFT_REPO_ROOT/bench.py --workload g1024 --steps 2 --warmup 1 --cpu-seconds 0 --no-prof > $GRAFT_REPO_ROOT/$OUT/pmc_g1024_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT/g1024 $OUT/r06_pmc_g1024_FETCH_WRITE.json "round 6 final sources" > $OUT/pmc_summary_g1024.log 2>&1
cp $OUT/r06_pmc_g1024_FETCH_WRITE.json profiles/ 2>/dev/null
rm -rf $OUT/g1024
timeout 400 python bench.py --workload g1024 --cpu-seconds 10 > $OUT/bench_g1024.json 2> $OUT/bench_g1024.err
# 6b. the ADA geometric block: one-kernel forward / adjoint against the four-pass composition, and the cost of aug=ada in the step (eager and captured; SGV_ADA_ADJOINT=0:
#     the differentiated calls run the composition, as in rounds 4-5)
timeout 300 python tools/ada_bench.py --static 0 > $OUT/r06_ada_bench_measured_margin.log 2>&1
timeout 300 python tools/ada_bench.py --static 1 > $OUT/r06_ada_bench_static_margin.log 2>&1
{
python tools/ada_step_bench.py --aug noaug 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada 2>&1 | tail -1
python tools/ada_step_bench.py --aug noaug --graphs 1 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada --graphs 1 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada --graphs 1 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada --graphs 1 --p 0.3 2>&1 | tail -1
} > $OUT/r06_ada_in_step.txt 2>&1; cat $OUT/r06_ada_in_step.txt
# 7. census of a captured main iteration
( cd /tmp && SGV_SELFTEST=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/census -- python $GRAFT_REPO_ROOT/tools/captured_census.py > /tmp/census.log 2>&1 )
f=$(find /tmp/census -name "*kernel_trace.csv" | head -1); python tools/captured_census_report.py $f > $OUT/r06_captured_census.txt 2>&1; head -3 $OUT/r06_captured_census.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_final/bench_*.json')):
    if f.endswith('_detail.json'):
        continue
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], len(json.dumps(d)), 'bytes |', round(d['value'], 1), d['unit'], round(d['ms_per_step'], 2), 'ms', '| mode', (d.get('config') or {}).get('headline_mode'), '| roofline frac', (d.get('roofline') or {}).get('frac'), 'traffic', (d.get('roofline') or {}).get('traffic'),
              '| ufd', (d.get('roofline_upfirdn2d') or {}).get('frac'), (d.get('roofline_upfirdn2d') or {}).get('traffic'), '| power', d.get('power'), '|', {k: v for k, v in d.items() if k.startswith('value_') and v is not None})
    except Exception as e:
        print(f, 'unreadable', e)
PY
