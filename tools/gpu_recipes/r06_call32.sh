#!/bin/bash
# call 32: adjoint v7 (plain-FMA gather over zero-initialised, guard-banded buffers) -- parity, timing
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c32
timeout 900 python -m pytest tests/test_augment.py -q -m gpu > gpurun_out/c32/pytest_augment.log 2>&1; tail -5 gpurun_out/c32/pytest_augment.log
timeout 300 python tools/ada_bench.py --static 0 > gpurun_out/c32/ada_bench_measured.log 2>&1; cat gpurun_out/c32/ada_bench_measured.log
timeout 300 python tools/ada_bench.py --static 1 > gpurun_out/c32/ada_bench_static.log 2>&1; cat gpurun_out/c32/ada_bench_static.log
