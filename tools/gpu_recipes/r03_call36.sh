# round 3, GPU call 36: BASELINE configs[4] record: rocprofv3 kernel stats and PMC FETCH / WRITE passes of `bench.py --workload g1024`, then the line itself (which reads them)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
C=${SGV_COMMIT:-0391c44}
B="python $GRAFT_REPO_ROOT/bench.py --workload g1024 --cpu-seconds 0"
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof36 -- $B --steps 16 --warmup 3 > /tmp/prof36.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof36 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_g1024_kernel_stats.csv
head -12 gpurun_out/r03_g1024_kernel_stats.csv | cut -c1-170
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc36/pmc_bench_$c -- $B --steps 2 --warmup 1 --no-prof > /tmp/pmc36_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py /tmp/pmc36 gpurun_out/r03_pmc_g1024_FETCH_WRITE.json $C
cp gpurun_out/r03_pmc_g1024_FETCH_WRITE.json profiles/
timeout 200 python bench.py --workload g1024 --cpu-seconds 10 2>/dev/null | tail -1 > gpurun_out/r03_bench_g1024.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_g1024.json')); print(round(d['value'],1), {k:d['roofline'][k] for k in ('achieved','frac','traffic','algorithmic_bytes_per_forward','launches_per_forward')})"
