# round 6, call 10: the 2x tile kernels and mode 2 IN the training step: eager step with per-launch events, tile forms on / off, same box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c10
mkdir -p $OUT
export TMPDIR=/tmp
OFF="--strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
for v in A B A2 B2; do
  case $v in A|A2) E="";; B|B2) E="SGV_UFD_TILE2X=0 SGV_UFD_TILE_EPI2=0";; esac
  env $E timeout 600 python bench.py --eager --cpu-seconds 0 $OFF --steps 10 --warmup 3 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  cp bench_detail.json $OUT/bench_${v}_detail.json
done
python - <<'PY'
import json
for v in ('A','B','A2','B2'):
    d=json.load(open(f'gpurun_out/r06_c10/bench_{v}_detail.json'))
    print(v, 'value', round(d['value'],1), 'ufd frac', d['roofline_upfirdn2d']['frac'], 'power', d.get('power'))
    kv=d['kernels_by_variant']
    print('   ', {k: (round(x['ms_per_step'],3), round(x['GBps'])) for k,x in kv.items() if k.startswith('ufd')})
    bs={round(r['algorithmic_MB']):(round(r['avg_us'],1), r['launches']) for r in d['upfirdn2d_by_size'][:16]}
    print('   ', bs)
PY
