cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 280 python bench.py --steps 16 --warmup 2 > gpurun_out/bench5.json 2> gpurun_out/bench5.err; grep -v amdgpu.ids gpurun_out/bench5.err | tail -8; cat gpurun_out/bench5.json
