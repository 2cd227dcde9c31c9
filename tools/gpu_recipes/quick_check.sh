# targeted GPU tests + the bench line without the CPU leg (about one GPU-minute)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_extras_gpu.py tests/test_networks.py -m gpu -q -x > gpurun_out/quick_tests.log 2>&1; grep -v amdgpu.ids gpurun_out/quick_tests.log | tail -3
timeout 100 python bench.py --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/quick_bench.json | cut -c1-200
