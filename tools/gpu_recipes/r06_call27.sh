#!/bin/bash
# call 27: adjoint kernel v4 (256 threads, 3 workgroups per CU) -- parity, time against the composition
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c27
timeout 900 python -m pytest tests/test_augment.py -q -m gpu > gpurun_out/c27/pytest_augment.log 2>&1; tail -30 gpurun_out/c27/pytest_augment.log
timeout 300 python tools/ada_bench.py --static 0 > gpurun_out/c27/ada_bench_measured.log 2>&1; cat gpurun_out/c27/ada_bench_measured.log
timeout 300 python tools/ada_bench.py --static 1 > gpurun_out/c27/ada_bench_static.log 2>&1; cat gpurun_out/c27/ada_bench_static.log
