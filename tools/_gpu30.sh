cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_conv_wrw_gpu.py tests/test_abi.py -m gpu -x -q -s > gpurun_out/t30_full.log 2>&1; grep -v amdgpu.ids gpurun_out/t30_full.log | grep "bf16x3 rel\|passed\|failed\|Error" | head -20
