# round 2, GPU call 6: full suite (fc v2, temporal encoder, ADA), bench with strict-fp32 and aug=ada companions
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
echo skip tests
SECONDS=0; timeout 600 python bench.py --steps 16 --warmup 3 2> gpurun_out/r02_b6.err | tail -1 > gpurun_out/r02_b6.json; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r02_b6.json')); print({k: d[k] for k in ('value','ms_per_step','value_strict_fp32','value_aug_ada')}); print(d['aug_ada']); print(d['strict_fp32']); print(d['cpu_baseline']); print({k:(round(v['ms_total']/d['steps'],2)) for k,v in d['kernels'].items()})"
echo "bench wall ${SECONDS}s"; grep -i "warm-up\|companion" gpurun_out/r02_b6.err | tail -8
