cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/t37_full.log 2>&1; grep -v amdgpu.ids gpurun_out/t37_full.log | tail -6
