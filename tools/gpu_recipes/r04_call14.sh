# round 4, GPU call 14: which operands still get a separate magnitude-bound pass (SGV_AMAX_TRACE) in the plain step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0 --no-prof"
SGV_AMAX_TRACE=1 timeout 300 python bench.py $OFF --steps 16 --warmup 16 > gpurun_out/r04_c14_trace.json 2> gpurun_out/r04_c14_trace.err; echo "rc=$?"
grep -A60 "amax trace" gpurun_out/r04_c14_trace.err | cut -c1-200
