cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/t13.log
timeout 200 python bench.py --steps 16 --warmup 2 --cpu-seconds 0 > gpurun_out/bench13.json 2> gpurun_out/bench13.err; grep -v amdgpu.ids gpurun_out/bench13.err | tail -3; cat gpurun_out/bench13.json
