#!/bin/bash
# call 36: per-sample time of the one-kernel adjoint / forward over 256 drawn maps: the slow ones
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c36
python tools/ada_adjoint_sweep.py --static 0 2>&1 | tee gpurun_out/c36/sweep_measured.txt
python tools/ada_adjoint_sweep.py --static 1 2>&1 | tee gpurun_out/c36/sweep_static.txt
