# round 2, GPU call 1: producer/consumer conv lab, full GPU suite (incl. the new bench-shape / mid-size / fused / stub tests), A/B bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 180 tools/conv_lab 5 ws > gpurun_out/r02_conv_lab_ws.log 2>&1; echo "lab rc=$?"; grep -v amdgpu.ids gpurun_out/r02_conv_lab_ws.log | tail -25
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r02_t1.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t1.log | tail -40
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 2> gpurun_out/r02_b1.err | tail -1 > gpurun_out/r02_b1.json; echo "bench rc=$?"; cut -c1-400 gpurun_out/r02_b1.json
SGV_CONV_WS=0 SGV_FUSED_CONV=0 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 2> gpurun_out/r02_b0.err | tail -1 > gpurun_out/r02_b0.json; cut -c1-400 gpurun_out/r02_b0.json
SGV_FUSED_CONV=0 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 2> gpurun_out/r02_b2.err | tail -1 > gpurun_out/r02_b2.json; cut -c1-400 gpurun_out/r02_b2.json
