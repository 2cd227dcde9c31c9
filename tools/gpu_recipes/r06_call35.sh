#!/bin/bash
# call 35: parameter slots for captured aug=ada phases; in-step cost of aug=ada, one-kernel adjoint against the composition (SGV_ADA_ADJOINT=0), eager and captured
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c35
timeout 900 python -m pytest tests/test_augment.py tests/test_extras_gpu.py tests/test_fc_gpu.py -q -m gpu -x 2>&1 | tail -3
{
python tools/ada_step_bench.py --aug noaug 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada --p 0.3 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada --p 0.3 2>&1 | tail -1
python tools/ada_step_bench.py --aug noaug --graphs 1 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada --graphs 1 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada --graphs 1 2>&1 | tail -1
python tools/ada_step_bench.py --aug ada --graphs 1 --p 0.3 2>&1 | tail -1
SGV_ADA_ADJOINT=0 python tools/ada_step_bench.py --aug ada --graphs 1 --p 0.3 2>&1 | tail -1
} | tee gpurun_out/c35/ada_in_step.txt
