# round 5, GPU call 21: call 20 sampled card0, which is not necessarily this job's GPU (the box shows several cards): the sampler now follows the HIP device's PCI address.
# The two speeds of the same graph against ITS card's clock / power; then the captured headline with the power sample in the line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 1 2; do SGV_SELFTEST=0 timeout 300 python tools/graph_sets_lab.py 3 0 > gpurun_out/r05_c21_clean_$r.log 2>&1; grep "^set\|^again\|Error" gpurun_out/r05_c21_clean_$r.log | cut -c1-900; done
SGV_SELFTEST=0 timeout 300 python tools/graph_sets_lab.py 2 1 > gpurun_out/r05_c21_stamped.log 2>&1; grep "^set\|^again\|Error" gpurun_out/r05_c21_stamped.log | cut -c1-900
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c21_captured.json 2> gpurun_out/r05_c21_captured.err; grep "per-iteration\|Error\|error" gpurun_out/r05_c21_captured.err | cut -c1-600; cat gpurun_out/r05_c21_captured.json | cut -c1-3000
timeout 300 python bench.py $OFF --eager --steps 20 --warmup 5 > gpurun_out/r05_c21_eager.json 2> gpurun_out/r05_c21_eager.err; grep "per-iteration\|Error\|error" gpurun_out/r05_c21_eager.err | cut -c1-600; python -c "
import json; d=json.loads(open('gpurun_out/r05_c21_eager.json').read()); print(d['value'], d.get('power'), d['roofline']['frac'])"
