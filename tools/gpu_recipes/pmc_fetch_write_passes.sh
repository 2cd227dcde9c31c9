cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_bench_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-prof > $GRAFT_REPO_ROOT/gpurun_out/pmc_bench_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, json
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    fs = glob.glob(f'gpurun_out/pmc_bench_{c}/*/*counter_collection.csv')
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        key = None
        for ns in ('(anonymous namespace)::', 'sgv_conv::', 'sgv_wrw::', 'sgv_gemm::'):
            if ns in k:
                key = k.split(ns)[1].split('(')[0][:80]
                break
        if key is None:
            continue
        agg[key][0] += 1; agg[key][1] += float(r['Counter_Value'])
    out[c] = {k: dict(launches=v[0], total_KB=v[1]) for k, v in agg.items()}
json.dump(out, open('gpurun_out/pmc_bench_summary_v2.json', 'w'), indent=1)
print({k: v for k, v in out['FETCH_SIZE'].items() if 'conv' in k or 'wrw' in k})
PY
rm -rf gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE
