# round 4, the record's last part (kernel sources unchanged since r04_final.sh: same digest, counter summary committed): whole GPU suite + smoke after the
# two test fixes, then the driver's command with profiles/r04_pmc_bench_step_FETCH_WRITE.json in place
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_final
mkdir -p $OUT/tables
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; print('csrc digest', c.source_digest())" | tee $OUT/digest.log
SGV_ERROR_TABLE_DIR=$OUT/tables timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_final/bench_driver_cmd.json') if l.startswith('{')][-1])
print(round(d['value'], 1), round(d['ms_per_step'], 2), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d['config'].items() if k.startswith('value_') or k.startswith('upfirdn')}, d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])
PY
