# round 4, addendum to the final record (kernel sources unchanged: the counter summary of r04_final.sh stays valid): the two test files that changed
# (inputs drawn on the device: tests reseed through TrainStep.reseed_inputs; networks_full at 8 x the reference's fp32 noise) and the driver's command
# with the companions on the headline's schedule mix (10 steps: one R1 iteration in ten)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_final
mkdir -p $OUT/tables
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; print('csrc digest', c.source_digest())"
SGV_ERROR_TABLE_DIR=$OUT/tables timeout 900 python -m pytest tests/test_extras_gpu.py tests/test_networks.py tests/test_augment.py -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_gpu_addendum.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_final/bench_driver_cmd.json') if l.startswith('{')][-1])
print(round(d['value'], 1), round(d['ms_per_step'], 2), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d['config'].items() if k.startswith('value_') or k.startswith('upfirdn')}, d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])
PY
