cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/t23.log
timeout 300 python tools/ops_bench.py --frames 32 --reps 20 --json gpurun_out/ops_bench_n32_v3.json 2>&1 | grep -v amdgpu.ids > gpurun_out/ops_bench_n32_v3.log; head -30 gpurun_out/ops_bench_n32_v3.log
timeout 300 python bench.py --steps 16 --warmup 2 > gpurun_out/bench23.json 2> gpurun_out/bench23.err; cat gpurun_out/bench23.json | cut -c1-300
