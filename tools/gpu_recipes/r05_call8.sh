# round 5, GPU call 8: where the eager step's host time goes (it is host-bound for ~9 of 158 ms per iteration; the captured step runs 149.7 ms) + the new captured-step companion of bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/cpu_step_profile.py 4 32 > gpurun_out/r05_c8_cpu_step_profile_b32.txt 2>&1; head -75 gpurun_out/r05_c8_cpu_step_profile_b32.txt | cut -c1-170
