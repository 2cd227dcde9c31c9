cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "upfirdn2d" 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/t10.log
for st in 8 16 32; do echo "== strip $st"; SGV_LANES_STRIP=$st timeout 120 python tools/ops_bench.py --frames 32 --reps 20 --only upfirdn2d 2>&1 | grep -v amdgpu.ids | head -9; done | tee gpurun_out/ops_bench_lanes.log
