cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_pointwise_gpu.py tests/test_networks.py tests/test_compat_reference.py tests/test_abi.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/t25.log
timeout 120 python tools/ops_bench.py --only pointwise --frames 96 --reps 10 --json gpurun_out/ops_pointwise.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ops25.log
timeout 200 python bench.py --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/b25.log
