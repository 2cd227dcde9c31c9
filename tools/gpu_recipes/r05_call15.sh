# round 5, GPU call 15: why the captured headline of call 14 (602 img/s) is not the standalone --graphs run (634-641): A = --graphs, B = captured headline without in-graph timing,
# C = with timestamp kernels, D = --graphs again (drift)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'graphs', d['config'].get('hip_graphs'), 'eager', d.get('value_eager'), 'roofline', (d.get('roofline') or {}).get('frac'))
PY
}
timeout 300 python bench.py $OFF --graphs --steps 20 --warmup 5 > gpurun_out/r05_c15_A.json 2> gpurun_out/r05_c15_A.err; show gpurun_out/r05_c15_A.json
SGV_BENCH_NO_STAMPS=1 timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c15_B.json 2> gpurun_out/r05_c15_B.err; show gpurun_out/r05_c15_B.json
timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c15_C.json 2> gpurun_out/r05_c15_C.err; show gpurun_out/r05_c15_C.json
timeout 300 python bench.py $OFF --graphs --steps 20 --warmup 5 > gpurun_out/r05_c15_D.json 2> gpurun_out/r05_c15_D.err; show gpurun_out/r05_c15_D.json
