cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_conv_wrw_gpu.py tests/test_conv3x3_gpu.py tests/test_conv2d_gradfix.py tests/test_networks.py tests/test_abi.py -m gpu -x -q -s > gpurun_out/t34_full.log 2>&1; grep -v amdgpu.ids gpurun_out/t34_full.log | grep "wrw-s2 rel\|passed\|failed\|Error\|error" | head -40
timeout 200 python bench.py --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/b34.log | cut -c1-400
