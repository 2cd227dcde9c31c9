cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof35 -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof35.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof35 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof35_kernel_stats.csv
rm -rf gpurun_out/prof35
grep '"metric"' gpurun_out/prof35.log | cut -c1-200
