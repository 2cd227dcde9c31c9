# round 5, GPU call 2: the W-stationary 1x1 kernel (tests, per-shape A/B against the tiled members) + where the 8-videos-per-GPU step (config 4's per-GPU workload) loses
# its time: per-variant tables at 8 and 32 videos, kernel traces (idle gaps) of the eager and the captured step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_extras_gpu.py -x -q -m gpu -k "conv1x1 or gemm" 2>&1 | tail -3
for n in 96 192; do for ws in 1 0; do
  echo "== gemm_bench N=$n SGV_GEMM_WSTAT=$ws"
  SGV_GEMM_BENCH_CHILD=1 N=$n SGV_GEMM_WSTAT=$ws timeout 200 python tools/gemm_bench.py > gpurun_out/r05_c2_gemm_n${n}_ws$ws.json 2> gpurun_out/r05_c2_gemm_n${n}_ws$ws.err
  python - gpurun_out/r05_c2_gemm_n${n}_ws$ws.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
for r in d['rows']:
    print('%-58s %-16s %7.3f ms %7.1f TF/s %7.1f GB/s' % (r['layer'], r['op'], r['ms'], r['TFLOPs'], r['GBps']))
print({k: v for k, v in d['variants'].items() if 'gemm' in k or 'wstat' in k})
PY
done; done
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0"
for b in 8 32; do
  timeout 300 python bench.py $OFF --batch-gpu $b --steps 20 --warmup 5 > gpurun_out/r05_c2_bench_b$b.json 2> gpurun_out/r05_c2_bench_b$b.err
  cp bench_detail.json gpurun_out/r05_c2_bench_detail_b$b.json
  tail -c 600 gpurun_out/r05_c2_bench_b$b.json; echo
done
for mode in eager graphs; do
  rm -rf /tmp/tr_$mode
  G=""; [ $mode = graphs ] && G="--graphs"
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$mode -- python $GRAFT_REPO_ROOT/bench.py $OFF --no-prof --batch-gpu 8 $G --steps 8 --warmup 4 > $GRAFT_REPO_ROOT/gpurun_out/r05_c2_trace_b8_$mode.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_c2_trace_b8_$mode.err )
  f=$(ls /tmp/tr_$mode/*/*kernel_trace.csv | head -1)
  echo "== b8 $mode: $(grep -o '"value":[0-9.]*' gpurun_out/r05_c2_trace_b8_$mode.json | head -1)"
  python tools/trace_gaps.py $f 5 | tee gpurun_out/r05_c2_gaps_b8_$mode.txt | head -12
  python - "$f" $mode <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d[r['Kernel_Name'].split('(')[0][-90:]][0] += 1
    d[r['Kernel_Name'].split('(')[0][-90:]][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
with open('gpurun_out/r05_c2_kernels_b8_%s.txt' % sys.argv[2], 'w') as fh:
    for k, (n, t) in sorted(d.items(), key=lambda kv: -kv[1][1]):
        fh.write('%10.3f ms %6d  %s\n' % (t, n, k))
PY
  gzip -c $f > gpurun_out/r05_c2_trace_b8_$mode.csv.gz; ls -la gpurun_out/r05_c2_trace_b8_$mode.csv.gz
done
