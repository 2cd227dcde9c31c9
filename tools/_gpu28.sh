cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_extras_gpu.py -m gpu -x -q -k "gemm or routes" 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/t28.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof28 -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof28.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof28 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof28_kernel_stats.csv
find gpurun_out/prof28 -type f ! -name "*stats*" -delete
tail -2 gpurun_out/prof28.log | cut -c1-300
