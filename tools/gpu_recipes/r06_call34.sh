#!/bin/bash
# call 34: cost of aug=ada in the step, one-kernel adjoint against the composition for the differentiated calls, eager and captured, p = 0 and p = 0.3; kernel trace
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c34
for rep in 1 2; do
python tools/ada_step_bench.py --aug noaug 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada 2>&1 | tail -1
done
python tools/ada_step_bench.py --aug ada --p 0.3 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada --p 0.3 2>&1 | tail -1
python tools/ada_step_bench.py --aug noaug --graphs 1 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada --graphs 1 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada --graphs 1 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c34/prof -o ada -- python $GRAFT_REPO_ROOT/tools/ada_step_bench.py --aug ada --steps 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/c34/prof -name '*kernel_stats.csv' | head -1); echo $f; head -40 "$f" | cut -c1-160; cp "$f" gpurun_out/c34/ada_step_kernel_stats.csv; rm -rf gpurun_out/c34/prof
