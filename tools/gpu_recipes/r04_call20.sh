# round 4, GPU call 20: persistent skip GEMM with two chunks of loads in flight (PF2) against one: tests, per-shape table (N = 96 and 192 frames), the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_extras_gpu.py tests/test_conv_f16split_gpu.py tests/test_networks.py -x -q -m gpu 2>&1 | tail -2
for n in 96 192; do for pf in 0 1; do
  N=$n SGV_GEMM_PF2=$pf SGV_GEMM_BENCH_CHILD=1 timeout 200 python tools/gemm_bench.py 2>/dev/null | tail -1 > gpurun_out/r04_c20_gemm_n${n}_pf$pf.json
done; done
python - <<'PY'
import json
for n in (96, 192):
    a = json.load(open('gpurun_out/r04_c20_gemm_n%d_pf0.json' % n))['rows']; b = json.load(open('gpurun_out/r04_c20_gemm_n%d_pf1.json' % n))['rows']
    for x, y in zip(a, b):
        print('N=%3d %-52s %-16s one chunk %.3f ms %6.0f GB/s | two %.3f ms %6.0f GB/s | x%.2f' % (n, x['layer'], x['op'], x['ms'], x['GBps'], y['ms'], y['GBps'], x['ms'] / y['ms']))
PY
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
for pf in 0 1; do
  SGV_GEMM_PF2=$pf timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r04_c20_bench_pf$pf.json 2> gpurun_out/r04_c20_bench_pf$pf.err
  python - $pf <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r04_c20_bench_pf%s.json' % sys.argv[1]) if l.startswith('{')][-1])
k = d['kernels_by_variant']
print('PF2', sys.argv[1], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'no_prof', round(d['value_no_prof'], 1), 'gemm stream ms', round(k['gemm_bf16x3_stream']['ms_per_step'], 2), round(k['gemm_bf16x3_stream']['GBps']))
PY
done
