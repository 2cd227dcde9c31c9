# round 5, GPU call 32: first-order demodulation-coefficient backward from the saved forward results (8 launches instead of 17 per layer), ToRGB's style gain on the
# dense kernel's output gain: the network / op parity tests, the count of small aten launches per eager main iteration, the captured headline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_extras_gpu.py tests/test_networks.py tests/test_train_step_gpu.py -q -m gpu -x 2>&1 | tail -4
SGV_SELFTEST=0 timeout 300 python tools/small_launch_sources.py 2>/dev/null | head -12 | cut -c1-260
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c32_captured.json 2> gpurun_out/r05_c32_captured.err; grep "per-iteration\|Error\|error" gpurun_out/r05_c32_captured.err | cut -c1-400; cut -c1-330 gpurun_out/r05_c32_captured.json; python -c "
import json; d=json.loads(open('gpurun_out/r05_c32_captured.json').read()); print(d['value'], d['value_eager'], d['roofline']['frac'], d['power'])"
