# round 3, GPU call 28: aten-level profile of two steps (who issues the element-wise launches) + kernel stats of the step on the current tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SGV_TORCH_PROFILE=gpurun_out/r03_torch_profile.txt timeout 300 python bench.py --cpu-seconds 0 --steps 4 --warmup 2 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --no-prof 2>/dev/null | cut -c1-100
timeout 300 python -m pytest tests/test_extras_gpu.py -m gpu -q -x -k "gemm" --timeout 200 2>&1 | grep -v amdgpu.ids | tail -2
