cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_conv_wrw_gpu.py tests/test_conv2d_gradfix.py tests/test_networks.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25 | tee gpurun_out/t29.log
timeout 200 python bench.py --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/b29.log
