cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/t41_full.log 2>&1; grep -v amdgpu.ids gpurun_out/t41_full.log | tail -3
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/smoke41.log
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
json.dump(out, open('gpurun_out/pmc_bench_summary_v3.json', 'w'), indent=1)
PY
rm -rf gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE
cp gpurun_out/pmc_bench_summary_v3.json profiles/r01_pmc_bench_step_v2_FETCH_WRITE.json
timeout 300 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/b41.json; cut -c1-200 gpurun_out/b41.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof41 -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof41.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof41 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof41_kernel_stats.csv
rm -rf gpurun_out/prof41
grep '"metric"' gpurun_out/prof41.log > gpurun_out/prof41_bench.json; cut -c1-160 gpurun_out/prof41_bench.json
