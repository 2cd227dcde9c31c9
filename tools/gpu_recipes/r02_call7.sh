# round 2, GPU call 7: rocprof kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) over the bench step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --no-prof"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02e -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02e.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_bench_$c -- $BENCH > $GRAFT_REPO_ROOT/gpurun_out/pmc_bench_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_bench_MFMA -- $BENCH > $GRAFT_REPO_ROOT/gpurun_out/pmc_bench_MFMA.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r02e -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_bench_step_kernel_stats_final.csv
grep '"metric"' gpurun_out/prof_r02e.log > gpurun_out/r02_bench_under_rocprof_final.json
python tools/pmc_summary.py gpurun_out gpurun_out/r02_pmc_bench_step_FETCH_WRITE.json "$SGV_COMMIT"
python tools/pmc_kernel_table.py gpurun_out/pmc_bench_MFMA > gpurun_out/r02_pmc_bench_step_MFMA_table.txt
rm -rf gpurun_out/prof_r02e gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE gpurun_out/pmc_bench_MFMA
head -30 gpurun_out/r02_bench_step_kernel_stats_final.csv | cut -c1-170
grep "conv\|wrw" gpurun_out/r02_pmc_bench_step_MFMA_table.txt | cut -c1-330
