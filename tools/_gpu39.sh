cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/t39_full.log 2>&1; grep -v amdgpu.ids gpurun_out/t39_full.log | tail -5
timeout 200 python bench.py --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/b39.json | cut -c1-200
