cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tools/g_forward_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g_forward.log
LOWP=bf16 timeout 120 python tools/g_forward_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/g_forward.log
timeout 200 python bench.py --steps 16 --warmup 2 --cpu-seconds 0 --lowp bf16 > gpurun_out/bench18_bf16.json 2> gpurun_out/bench18.err; grep -v amdgpu.ids gpurun_out/bench18.err | tail -3; cat gpurun_out/bench18_bf16.json
timeout 200 python bench.py --steps 16 --warmup 2 --cpu-seconds 0 --lowp fp16 > gpurun_out/bench18_fp16.json 2> gpurun_out/bench18b.err; grep -v amdgpu.ids gpurun_out/bench18b.err | tail -3; cat gpurun_out/bench18_fp16.json | cut -c1-400
