cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/t43_full.log 2>&1; grep -v amdgpu.ids gpurun_out/t43_full.log | tail -2
timeout 300 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/b43.json; cut -c1-200 gpurun_out/b43.json
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof43 -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --steps 16 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof43.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof43 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof43_kernel_stats.csv
rm -rf gpurun_out/prof43
grep '"metric"' gpurun_out/prof43.log > gpurun_out/prof43_bench.json; cut -c1-160 gpurun_out/prof43_bench.json
