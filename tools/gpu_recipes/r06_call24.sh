#!/bin/bash
# call 24: adjoint kernel -- parity, then a step-by-step ablation (SGV_ADJ_LAB bits skip steps) and a kernel trace
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c24
timeout 900 python -m pytest tests/test_augment.py -q -m gpu > gpurun_out/c24/pytest_augment.log 2>&1; tail -30 gpurun_out/c24/pytest_augment.log
for lab in 0 64 65 72 73 75 79 95 127; do
  echo "== SGV_ADJ_LAB=$lab"; SGV_ADJ_LAB=$lab timeout 300 python tools/ada_bench.py --static 0 --rounds 3 2>&1 | grep "one kernel backward"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c24/prof -o ada -- python $GRAFT_REPO_ROOT/tools/ada_bench.py --static 0 --rounds 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/c24/prof -name '*kernel_stats.csv' | head -1); head -12 "$f" | cut -c1-200
