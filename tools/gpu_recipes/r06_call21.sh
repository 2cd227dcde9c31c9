# round 6, call 21: where the skip-branch 1x1 products stand per shape and op (forward / data gradient / weight gradient), 96 and 192 frames
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c21
mkdir -p $OUT
export TMPDIR=/tmp
for n in 96 192; do
  N=$n SGV_GEMM_BENCH_CHILD=1 timeout 300 python tools/gemm_bench.py > $OUT/gemm_bench_n$n.json 2> $OUT/err_$n.log
  python - <<PY
import json
d=json.loads(open('gpurun_out/r06_c21/gemm_bench_n$n.json').read().strip().splitlines()[-1])
print('N=$n', d['mode'])
for r in d['rows']:
    print(f"  {r['layer']:48s} {r['op']:16s} {r['ms']*1e3:8.1f} us  {r['TFLOPs']:6.1f} TF  {r['GBps']:7.0f} GB/s  launches {r['launches']}")
PY
done
