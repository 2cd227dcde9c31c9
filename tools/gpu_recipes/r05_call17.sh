# round 5, GPU call 17b: second capture inside warm-up iteration 1, a full R1 iteration behind it; per-iteration device times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c17_a.json 2> gpurun_out/r05_c17_a.err; grep "per-iteration\|Error\|error" gpurun_out/r05_c17_a.err | cut -c1-600; cut -c1-400 gpurun_out/r05_c17_a.json
SGV_BENCH_NO_SWAP=1 timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c17_b.json 2> gpurun_out/r05_c17_b.err; grep "per-iteration\|Error\|error" gpurun_out/r05_c17_b.err | cut -c1-600; cut -c1-400 gpurun_out/r05_c17_b.json
