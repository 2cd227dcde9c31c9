# round 4, GPU call 17: synthetic clips / latents drawn on the device (no 19-MB pageable upload + sync per iteration): the step, eager and captured
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0"
timeout 400 python bench.py $OFF --ada-steps 12 --steps 20 --warmup 5 > gpurun_out/r04_c17_bench.json 2> gpurun_out/r04_c17_bench.err; echo "rc=$?"
timeout 400 python bench.py $OFF --ada-steps 0 --graphs --steps 20 --warmup 5 > gpurun_out/r04_c17_bench_graphs.json 2> gpurun_out/r04_c17_bench_graphs.err; echo "rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/r04_c17_bench.json', 'gpurun_out/r04_c17_bench_graphs.json'):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f, 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'no_prof', d.get('value_no_prof'), 'ada', (d.get('aug_ada') or {}).get('value'))
PY
