set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/t2.log
timeout 900 python bench.py --steps 16 --warmup 3 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; tail -3 gpurun_out/bench2.err; cat gpurun_out/bench2.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof2 -name "*stats*" | head
