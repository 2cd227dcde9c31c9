#!/bin/bash
# call 29: adjoint v5 (dy box prefetched across channels): 256-thread build (A) against 384-thread build (B, in tree), parity on B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c29
timeout 900 python -m pytest tests/test_augment.py -q -m gpu > gpurun_out/c29/pytest_augment.log 2>&1; tail -5 gpurun_out/c29/pytest_augment.log
for lib in gpurun_out/libsgv_A256.so stylegan-v_amd/csrc/libsgv_hip.so gpurun_out/libsgv_A256.so stylegan-v_amd/csrc/libsgv_hip.so; do
  echo "== $lib"; SGV_LIB_PATH=$PWD/$lib timeout 300 python tools/ada_bench.py --static 0 --rounds 3 2>&1 | grep "one kernel backward\|^#"
done
