cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_extras_gpu.py tests/test_pointwise_gpu.py tests/test_networks.py tests/test_conv2d_gradfix.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/t27.log
timeout 120 python tools/ops_bench.py --only gemm --frames 96 --reps 10 --json gpurun_out/ops_gemm.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ops27.log
timeout 200 python bench.py --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/b27.log
