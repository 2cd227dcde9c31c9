# round 2, GPU call 12: kernel stats of the bench step (stride-2 producer / consumer kernels, fc edges)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02c -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02c.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r02c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_bench_step_kernel_stats_v3.csv
grep '"metric"' gpurun_out/prof_r02c.log > gpurun_out/r02_bench_under_rocprof_v3.json
rm -rf gpurun_out/prof_r02c
head -24 gpurun_out/r02_bench_step_kernel_stats_v3.csv | cut -c1-150
