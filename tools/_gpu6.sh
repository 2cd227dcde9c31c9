cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 260 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof6 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof6.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof6.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof6 -type f | head; f=$(find gpurun_out/prof6 -name "*kernel_stats.csv" | head -1); head -40 $f
# keep only the small summaries (the raw trace is large)
find gpurun_out/prof6 -name "*kernel_trace.csv" -size +20M -delete
