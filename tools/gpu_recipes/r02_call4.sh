# round 2, GPU call 4: fused test fix, rocprof kernel stats of the bench step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_fused_conv_gpu.py -m gpu -q --timeout 300 > gpurun_out/r02_t4.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t4.log | tail -4
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02a -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --cpu-seconds 0 --strict-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02a.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r02a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_bench_step_kernel_stats_v1.csv
rm -rf gpurun_out/prof_r02a
grep '"metric"' gpurun_out/prof_r02a.log > gpurun_out/r02_bench_under_rocprof_v1.json; cut -c1-200 gpurun_out/r02_bench_under_rocprof_v1.json
head -45 gpurun_out/r02_bench_step_kernel_stats_v1.csv | cut -c1-200
