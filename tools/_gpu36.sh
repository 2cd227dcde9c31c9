cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t36_full.log 2>&1; grep -v amdgpu.ids gpurun_out/t36_full.log | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/smoke36.log
timeout 300 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/b36.json; cut -c1-300 gpurun_out/b36.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof36 -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof36.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof36 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof36_kernel_stats.csv
rm -rf gpurun_out/prof36
grep '"metric"' gpurun_out/prof36.log > gpurun_out/prof36_bench.json; cut -c1-200 gpurun_out/prof36_bench.json
