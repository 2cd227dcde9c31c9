cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export MIOPEN_FIND_MODE=2
timeout 170 python tools/conv_probe.py 96 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_probe_imm.log
