# round 4, GPU call 9: gather-form resample adjoint (tests + the aug=ada step), the driver's command with every companion, hipGraph replay at 32 and 8 videos/GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 300 python -m pytest tests/test_augment.py tests/test_conv_f16split_gpu.py -m gpu -q --timeout 200 2>&1 | grep -v amdgpu.ids | tail -4
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_c9_bench_driver_cmd.json 2> gpurun_out/r04_c9_bench_driver_cmd.err ) 2>&1 | grep real
Q="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0"
timeout 240 python bench.py $Q --steps 12 --warmup 3 --graphs > gpurun_out/r04_c9_bench_graphs.json 2> gpurun_out/r04_c9_bench_graphs.err; echo "graphs rc=$?"
timeout 240 python bench.py $Q --steps 16 --warmup 3 --batch-gpu 8 --graphs > gpurun_out/r04_c9_bench_batch8_graphs.json 2> gpurun_out/r04_c9_bench_batch8_graphs.err; echo "batch8 graphs rc=$?"
timeout 240 python bench.py $Q --steps 16 --warmup 3 --batch-gpu 8 > gpurun_out/r04_c9_bench_batch8_eager.json 2> gpurun_out/r04_c9_bench_batch8_eager.err; echo "batch8 eager rc=$?"
timeout 240 python bench.py $Q --steps 16 --warmup 3 --batch-gpu 8 --graphs --lowp bf16 > gpurun_out/r04_c9_bench_batch8_bf16_graphs.json 2> gpurun_out/r04_c9_bench_batch8_bf16_graphs.err; echo "batch8 bf16 graphs rc=$?"
python - <<'PY'
import json
def line(name):
    try:
        return json.loads(open(f'gpurun_out/{name}.json').read().strip().splitlines()[-1])
    except Exception as e:
        print(name, 'no line', e); return None
d = line('r04_c9_bench_driver_cmd')
if d:
    print('driver cmd: value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 1), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d['config'].items() if k.startswith('value_') or k.startswith('upfirdn2d')})
    print('  roofline', {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'traffic')}, '| cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'cores', 'main_iteration_samples', 'seconds_reg_iteration')})
for name in ('r04_c9_bench_graphs', 'r04_c9_bench_batch8_graphs', 'r04_c9_bench_batch8_eager', 'r04_c9_bench_batch8_bf16_graphs'):
    d = line(name)
    if d: print(name, 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2))
PY
