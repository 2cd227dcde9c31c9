# round 3, GPU call 15: fused-layer 16-bit tests after the sign-pattern-aware gradient check; --lowp with hipGraphs (the 16-bit step is launch-bound)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_lowp_gpu.py -m gpu -q -s --timeout 500 -k "fused" > gpurun_out/r03_t15.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t15.log | grep -E "passed|failed|FAILED|Error|rel-L2" | cut -c1-300 | tail -20
for mode in "--lowp bf16 --graphs" "--lowp fp16 --graphs" "--graphs"; do
tag=$(echo $mode | tr -d ' -')
timeout 400 python bench.py --cpu-seconds 0 $mode --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 2> gpurun_out/r03_b15_$tag.err | tail -1 > gpurun_out/r03_b15_$tag.json; echo "bench $mode rc=$?"; cut -c1-220 gpurun_out/r03_b15_$tag.json; tail -2 gpurun_out/r03_b15_$tag.err
done
