# round 2, GPU call 14: full GPU suite + generator-only forward lines (configs[1] and the per-GPU share of configs[4])
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r02_t14.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t14.log | tail -8
timeout 300 python tools/g_forward_bench.py 2>/dev/null | tail -1 > gpurun_out/r02_g_forward_256.json; cat gpurun_out/r02_g_forward_256.json
RES=1024 F=16 B=1 ITERS=10 timeout 600 python tools/g_forward_bench.py 2> gpurun_out/r02_g1024.err | tail -1 > gpurun_out/r02_g_forward_1024x16.json; cat gpurun_out/r02_g_forward_1024x16.json; tail -3 gpurun_out/r02_g1024.err | cut -c1-300
