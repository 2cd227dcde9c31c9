#!/bin/bash
# call 41: where the small aten launches of a main iteration come from (current tree)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c41
timeout 600 python tools/small_launch_sources.py > gpurun_out/c41/small_launch_sources.txt 2>&1; head -70 gpurun_out/c41/small_launch_sources.txt | cut -c1-250
