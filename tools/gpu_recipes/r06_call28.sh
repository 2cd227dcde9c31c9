#!/bin/bash
# call 28: step ablation of adjoint v4 (SGV_ADJ_LAB bits: 1 a, 2 b1, 4 b2, 8 c, 16 d1, 32 d2, 64 build)
cd "$GRAFT_REPO_ROOT"
for lab in 0 64 72 88 120 124 126 127 8 4 2 1 16; do
  echo "== SGV_ADJ_LAB=$lab"; SGV_ADJ_LAB=$lab timeout 300 python tools/ada_bench.py --static 0 --rounds 3 2>&1 | grep "one kernel backward" | head -1
done
