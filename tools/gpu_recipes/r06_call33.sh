#!/bin/bash
# call 33: whole -m gpu suite on the tree with the one-kernel ADA adjoint, then bench.py without flags (aug=ada companions: eager and captured)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c33
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/c33/pytest_gpu.log 2>&1; tail -6 gpurun_out/c33/pytest_gpu.log
timeout 1200 python bench.py > gpurun_out/c33/bench.json 2> gpurun_out/c33/bench.log; tail -3 gpurun_out/c33/bench.log; python - <<'PY'
import json
d = json.loads(open('gpurun_out/c33/bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step')}, d['config'].get('headline_mode'), {k: v for k, v in d['config'].items() if k.startswith('value_')})
PY
