cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_extras_gpu.py tests/test_networks.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/t12.log
timeout 200 python tools/ops_bench.py --frames 32 --reps 20 --json gpurun_out/ops_bench_n32_v2.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ops_bench_n32_v2.log
