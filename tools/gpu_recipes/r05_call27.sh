# round 5, GPU call 27: the stride-1 convolution writes its own timestamps inside a capture (no bracketing kernels): new test, the convolution / graph tests, the
# captured headline with ONE graph set carrying the timing in every replay, and the same under the tracer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; print('csrc digest', c.source_digest()); sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 900 python -m pytest tests/test_extras_gpu.py tests/test_abi.py tests/test_conv3x3_gpu.py tests/test_conv_f16split_gpu.py -q -m gpu -x 2>&1 | tail -5
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c27_captured.json 2> gpurun_out/r05_c27_captured.err; grep "per-iteration\|Error\|error" gpurun_out/r05_c27_captured.err | cut -c1-500; cut -c1-2600 gpurun_out/r05_c27_captured.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_c27_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $OFF > $GRAFT_REPO_ROOT/gpurun_out/r05_c27_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_c27_under_rocprof.err )
find gpurun_out/r05_c27_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_c27_kernel_stats.csv; rm -rf gpurun_out/r05_c27_stats
grep "per-iteration" gpurun_out/r05_c27_under_rocprof.err | cut -c1-400
python - <<'PY'
import json, csv
d = json.loads(open('gpurun_out/r05_c27_under_rocprof.json').read())
print('line under the tracer:', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['launches'])
rows = list(csv.DictReader(open('gpurun_out/r05_c27_kernel_stats.csv')))
tot = n = 0
for r in rows:
    if 'conv3x3_ws_kernel' in r['Name'][:40]:
        tot += float(r['TotalDurationNs']); n += int(r['Calls'])
print('tracer conv3x3_ws avg us', tot / n / 1e3, n)
print([r['Name'][:40] for r in rows if 'stamp' in r['Name']])
PY
