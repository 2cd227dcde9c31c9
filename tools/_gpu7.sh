cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python bench.py --steps 16 --warmup 2 --cpu-seconds 0 > gpurun_out/bench7.json 2> gpurun_out/bench7.err; grep -v amdgpu.ids gpurun_out/bench7.err | tail -5; cat gpurun_out/bench7.json
