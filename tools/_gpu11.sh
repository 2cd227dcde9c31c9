cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tools/ops_bench.py --frames 32 --reps 30 --only upfirdn2d 2>&1 | grep -v amdgpu.ids | head -5 | tee gpurun_out/ops_bench_lanes2.log
timeout 120 python tools/ops_bench.py --frames 96 --reps 10 --only upfirdn2d 2>&1 | grep -v amdgpu.ids | head -5 | tee -a gpurun_out/ops_bench_lanes2.log
