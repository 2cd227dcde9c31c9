set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -5
nproc; lscpu | grep "Model name"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_extras_gpu.py -m gpu -x -q -k "not reference_time_encoder" 2>&1 | tail -25 > gpurun_out/t1.log
cat gpurun_out/t1.log
timeout 600 python tools/ops_bench.py --frames 32 --reps 20 --json gpurun_out/ops_bench_n32.json 2>&1 | tee gpurun_out/ops_bench_n32.log
timeout 600 python tools/ops_bench.py --frames 96 --reps 10 --only upfirdn2d --json gpurun_out/ops_bench_n96.json 2>&1 | tee gpurun_out/ops_bench_n96.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python $GRAFT_REPO_ROOT/tools/ops_bench.py --frames 32 --reps 10 --only upfirdn2d > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head
