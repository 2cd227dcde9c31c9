# round 4, GPU call 16: torch's own element-wise / copy / reduction launches of two plain iterations, by input shape
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SGV_TORCH_PROFILE=gpurun_out/r04_c16_torch_profile_plain.txt SGV_TORCH_PROFILE_FROM=1 timeout 240 python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --steps 2 --warmup 3 --no-prof > /dev/null 2> gpurun_out/r04_c16.err; echo "rc=$?"
grep -A62 "^torch op (2 steps)" gpurun_out/r04_c16_torch_profile_plain.txt | cut -c1-170
