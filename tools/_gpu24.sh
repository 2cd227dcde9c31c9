cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/t24.log
