# round 4, GPU call 23: split-K for the 16^2 / 8^2 kernel when a launch has fewer tiles than half the CUs: tests, then 8 and 32 videos per GPU against SGV_CONV_SMALL_KSPLIT=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv3x3_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_networks.py tests/test_conv_f16split_gpu.py -x -q -m gpu 2>&1 | tail -2
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
for ks in 1 0; do for b in 8 32; do
  SGV_CONV_SMALL_KSPLIT=$ks timeout 300 python bench.py $OFF --batch-gpu $b --steps 20 --warmup 5 > gpurun_out/r04_c23_bench_ks${ks}_b$b.json 2> gpurun_out/r04_c23_bench_ks${ks}_b$b.err
  python - $ks $b <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r04_c23_bench_ks%s_b%s.json' % (sys.argv[1], sys.argv[2])) if l.startswith('{')][-1])
k = d['kernels_by_variant']['conv_small']
print('ksplit', 'off' if sys.argv[1] == '1' else 'auto', 'batch', sys.argv[2], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'no_prof', round(d['value_no_prof'], 1), 'conv_small ms', round(k['ms_per_step'], 2), 'frac', round(k['frac_of_ceiling'], 3))
PY
done; done
