# round 5, GPU call 11 (re-run as call 14 with timestamp kernels instead of event nodes): the captured step as the one-GPU headline of bench.py with per-launch timing inside the graphs
# does hipEventElapsedTime read them after a replay, do the roofline objects agree with the eager run's, what does the line say
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c11_captured.json 2> gpurun_out/r05_c11_captured.err; echo "captured rc=$?"; tail -c 2300 gpurun_out/r05_c11_captured.json; echo; grep -i "error\|traceback" gpurun_out/r05_c11_captured.err | head -5
timeout 300 python bench.py $OFF --eager --steps 20 --warmup 5 > gpurun_out/r05_c11_eager.json 2> gpurun_out/r05_c11_eager.err; echo "eager rc=$?"; tail -c 1500 gpurun_out/r05_c11_eager.json; echo
timeout 600 python -m pytest tests/test_extras_gpu.py tests/test_abi.py -q -m gpu -k "graph or capture or abi or prof" 2>&1 | tail -3
