# round 4, GPU call 18: the adjoint-gather test that failed once in the addendum run: five repetitions with full output
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  timeout 300 python -m pytest tests/test_augment.py -q -m gpu -k "adjoint_gather" 2>&1 | grep -v amdgpu.ids | tail -25
done > gpurun_out/r04_c18_adjoint_repeats.log 2>&1
grep -c "1 passed" gpurun_out/r04_c18_adjoint_repeats.log; grep -B2 -A18 "Error\|assert" gpurun_out/r04_c18_adjoint_repeats.log | head -60
