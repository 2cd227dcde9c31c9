cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 260 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof8 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof8.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof8.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof8 -type f | head; f=$(find gpurun_out/prof8 -name "*kernel_stats.csv" | head -1); head -60 $f | cut -c1-200
# keep only the small summaries (the raw trace is large)
find gpurun_out/prof8 -name "*kernel_trace.csv" -size +20M -delete
