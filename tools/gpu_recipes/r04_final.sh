# round 4, the record on the final kernel sources: full GPU suite + smoke, counter passes (FETCH_SIZE / WRITE_SIZE, separate runs) -> digest-stamped summary,
# the driver's bench command with that summary in place, rocprofv3 --kernel-trace --stats of the step, the 1024^2 synthesis workload, 8 videos / GPU lines
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_final
mkdir -p $OUT/tables
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; print('csrc digest', c.source_digest()); sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
OFF="--strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0"
# 1. tests + smoke
SGV_ERROR_TABLE_DIR=$OUT/tables timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# 2. counter passes (own runs: --pmc with --kernel-trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_bench_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-prof $OFF > $GRAFT_REPO_ROOT/$OUT/pmc_bench_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT $OUT/r04_pmc_bench_step_FETCH_WRITE.json "39d8a60" > $OUT/pmc_summary.log 2>&1; tail -2 $OUT/pmc_summary.log | cut -c1-300
cp $OUT/r04_pmc_bench_step_FETCH_WRITE.json profiles/
rm -rf $OUT/pmc_bench_FETCH_SIZE $OUT/pmc_bench_WRITE_SIZE
# 3. the driver's command
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
# 4. kernel stats of the step (same workload, companions off so that the table is the headline step's)
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 $OFF > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.err )
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r04_bench_step_kernel_stats_final.csv
rm -rf $OUT/stats
# 5. aug=ada step kernel stats (the fused geometric kernel in context)
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_ada -- python $GRAFT_REPO_ROOT/bench.py --aug ada --steps 12 --warmup 4 --cpu-seconds 0 --no-prof $OFF > $GRAFT_REPO_ROOT/$OUT/bench_ada_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_ada_under_rocprof.err )
find $OUT/stats_ada -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r04_ada_step_kernel_stats_final.csv
rm -rf $OUT/stats_ada
# 6. other configurations
timeout 300 python bench.py --graphs --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_graphs.json 2> $OUT/bench_graphs.err
timeout 300 python bench.py --batch-gpu 8 --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_batch8_eager.json 2> $OUT/bench_batch8_eager.err
timeout 300 python bench.py --batch-gpu 8 --graphs --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_batch8_graphs.json 2> $OUT/bench_batch8_graphs.err
SGV_CONV_TERMS=1 SGV_WRW_TERMS=1 timeout 300 python bench.py --batch-gpu 8 --graphs --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_batch8_bf16_graphs.json 2> $OUT/bench_batch8_bf16_graphs.err
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/g1024/pmc_bench_$c -- python $GRAFT_REPO_ROOT/bench.py --workload g1024 --steps 2 --warmup 1 --cpu-seconds 0 --no-prof > $GRAFT_REPO_ROOT/$OUT/pmc_g1024_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT/g1024 $OUT/r04_pmc_g1024_FETCH_WRITE.json "39d8a60" > $OUT/pmc_summary_g1024.log 2>&1
cp $OUT/r04_pmc_g1024_FETCH_WRITE.json profiles/ 2>/dev/null
rm -rf $OUT/g1024
timeout 400 python bench.py --workload g1024 --cpu-seconds 10 > $OUT/bench_g1024.json 2> $OUT/bench_g1024.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_final/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], round(d['value'], 1), d['unit'], round(d['ms_per_step'], 2), 'ms', '| roofline frac', (d.get('roofline') or {}).get('frac'), 'traffic', (d.get('roofline') or {}).get('traffic'), '| cfg', {k: v for k, v in d['config'].items() if k.startswith('value_') or k in ('hip_graphs', 'videos_per_gpu')})
    except Exception as e:
        print(f, 'unreadable', e)
PY
