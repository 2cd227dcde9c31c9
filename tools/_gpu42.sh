cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_conv3x3_gpu.py tests/test_networks.py tests/test_conv2d_gradfix.py -m gpu -q -x > gpurun_out/t42_full.log 2>&1; grep -v amdgpu.ids gpurun_out/t42_full.log | tail -3
timeout 200 python bench.py --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/b42.json | cut -c1-200
