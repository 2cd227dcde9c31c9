# round 4, GPU call 7: the step with Dmain as one pass (new default) + inherited bounds; aten-level profile of two steps (who issues the ~1,100 torch launches);
# remaining separate bound passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 200 python -m pytest tests/test_conv_f16split_gpu.py -m gpu -q --timeout 200 2>&1 | grep -v amdgpu.ids | tail -3
B="python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --steps 12 --warmup 3"
timeout 240 $B > gpurun_out/r04_c7_bench.json 2> gpurun_out/r04_c7_bench.err; echo "bench rc=$?"
SGV_D_CONCAT=0 timeout 240 $B > gpurun_out/r04_c7_bench_two_pass.json 2> gpurun_out/r04_c7_bench_two_pass.err; echo "bench two-pass rc=$?"
SGV_AMAX_TRACE=1 SGV_TORCH_PROFILE=gpurun_out/r04_c7_torch_profile.txt timeout 240 python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --steps 2 --warmup 2 --no-prof > /dev/null 2> gpurun_out/r04_c7_trace.err; echo "profile rc=$?"
grep -A 14 "amax trace" gpurun_out/r04_c7_trace.err | cut -c1-180
python - <<'PY'
import json
for name in ('r04_c7_bench', 'r04_c7_bench_two_pass'):
    d = json.loads(open(f'gpurun_out/{name}.json').read().strip().splitlines()[-1])
    print(name, 'value', round(d['value'], 1), 'no_prof', round(d['value_no_prof'] or 0, 1), 'ms', round(d['ms_per_step'], 1), 'launches/step', d['config']['native_launches_per_step'])
    kv = d.get('kernels_by_variant') or {}
    print('   native ms/step', round(sum(v['ms_per_step'] for v in kv.values()), 1), 'absmax', (d.get('kernels') or {}).get('absmax'))
PY
sed -n '/by call count/,$p' gpurun_out/r04_c7_torch_profile.txt | awk 'NR<=70' | cut -c1-150
