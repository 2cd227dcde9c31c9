# round 5, GPU call 24: concurrency of a matrix-bound and an HBM-bound kernel of the step on two streams (tools/overlap_lab.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
timeout 200 python tools/overlap_lab.py > gpurun_out/r05_c24_overlap.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_c24_overlap.log | tail -12 | cut -c1-400
