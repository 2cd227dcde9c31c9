# round 4, GPU call 6: the whole -m gpu suite under the new default arithmetic (terms = 4), the trace of the bound passes that are still separate, and
# the train-step tests with the Dmain phase as one discriminator pass (SGV_D_CONCAT=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_tables
export TMPDIR=/tmp SGV_ERROR_TABLE_DIR=$GRAFT_REPO_ROOT/gpurun_out/r04_tables
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 900 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r04_c6_all_tests.log 2>&1; echo "all tests rc=$?"
grep -v amdgpu.ids gpurun_out/r04_c6_all_tests.log | grep -E "passed|failed|FAILED|Error" | cut -c1-300 | tail -30
SGV_AMAX_TRACE=1 timeout 200 python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --steps 4 --warmup 2 --no-prof > gpurun_out/r04_c6_trace.json 2> gpurun_out/r04_c6_trace.err; echo "trace rc=$?"
grep -A 45 "amax trace" gpurun_out/r04_c6_trace.err | cut -c1-200
SGV_D_CONCAT=1 timeout 400 python -m pytest tests/test_extras_gpu.py tests/test_networks.py tests/test_augment.py -m gpu -q --timeout 300 > gpurun_out/r04_c6_dconcat_tests.log 2>&1; echo "dconcat tests rc=$?"
grep -v amdgpu.ids gpurun_out/r04_c6_dconcat_tests.log | grep -E "passed|failed|FAILED|Error" | cut -c1-300 | tail -10
