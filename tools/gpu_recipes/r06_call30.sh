#!/bin/bash
# call 30: adjoint lab builds (lab_libs/, SGV_LIB_PATH): threads 256 / 384 x dy-box prefetch on / off, same box
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for name in A256 B384 C256np D384np; do
  echo "== $name"; SGV_LIB_PATH=$PWD/lab_libs/libsgv_$name.so timeout 300 python tools/ada_bench.py --static 0 --rounds 3 2>&1 | grep "one kernel backward"
done; done
