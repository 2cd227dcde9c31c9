#!/bin/bash
# call 23: the ADA geometric block's adjoint as one kernel -- parity tests, then its time against the composition's backward
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c23
timeout 900 python -m pytest tests/test_augment.py -x -q -m gpu > gpurun_out/c23/pytest_augment.log 2>&1; tail -15 gpurun_out/c23/pytest_augment.log
timeout 300 python tools/ada_bench.py > gpurun_out/c23/ada_bench_static.log 2>&1; cat gpurun_out/c23/ada_bench_static.log
timeout 300 python tools/ada_bench.py --static 0 > gpurun_out/c23/ada_bench_measured.log 2>&1; cat gpurun_out/c23/ada_bench_measured.log
