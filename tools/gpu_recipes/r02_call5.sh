# round 2, GPU call 5: fc kernel, graph replay, wrw ws in the product, bench at 32 and 8 videos per GPU (eager vs hipGraph)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/r02_t5.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t5.log | tail -12
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 2> gpurun_out/r02_b5.err | tail -1 > gpurun_out/r02_b5.json; echo "bench rc=$?"; cut -c1-200 gpurun_out/r02_b5.json
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --batch-gpu 8 2> gpurun_out/r02_b5_b8.err | tail -1 > gpurun_out/r02_b5_b8.json; cut -c1-200 gpurun_out/r02_b5_b8.json
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --batch-gpu 8 --graphs 2> gpurun_out/r02_b5_b8g.err | tail -1 > gpurun_out/r02_b5_b8g.json; cut -c1-200 gpurun_out/r02_b5_b8g.json; tail -3 gpurun_out/r02_b5_b8g.err
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --graphs 2> gpurun_out/r02_b5_g.err | tail -1 > gpurun_out/r02_b5_g.json; cut -c1-200 gpurun_out/r02_b5_g.json
