#!/bin/bash
# call 37: forward kernel with 8x8 / 4x4 sub-tiles for zoom-out maps: parity; time against the previous build (fwd0), with (tree) and without (fwdA) the 128-register bound
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c37
timeout 900 python -m pytest tests/test_augment.py -q -m gpu -x 2>&1 | tail -3
for lib in lab_libs/libsgv_fwd0.so lab_libs/libsgv_fwdA.so stylegan-v_amd/csrc/libsgv_hip.so lab_libs/libsgv_fwd0.so lab_libs/libsgv_fwdA.so stylegan-v_amd/csrc/libsgv_hip.so; do
  echo "== $lib"; SGV_LIB_PATH=$PWD/$lib timeout 300 python tools/ada_bench.py --static 0 --rounds 3 2>&1 | grep "one kernel forward"
  SGV_LIB_PATH=$PWD/$lib timeout 300 python tools/ada_bench.py --static 1 --rounds 3 2>&1 | grep "one kernel forward"
done
