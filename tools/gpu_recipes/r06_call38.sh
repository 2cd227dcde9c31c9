#!/bin/bash
# call 38: forward as two kernels (16-tiles / sub-tiles): parity, ada_bench, then the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c38
timeout 900 python -m pytest tests/test_augment.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/ada_bench.py --static 0 > gpurun_out/c38/ada_bench_measured.log 2>&1; cat gpurun_out/c38/ada_bench_measured.log
timeout 300 python tools/ada_bench.py --static 1 > gpurun_out/c38/ada_bench_static.log 2>&1; cat gpurun_out/c38/ada_bench_static.log
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/c38/pytest_gpu.log 2>&1; tail -6 gpurun_out/c38/pytest_gpu.log
