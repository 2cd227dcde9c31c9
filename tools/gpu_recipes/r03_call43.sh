# round 3, GPU call 43: the module-level goldens and the training-step tests on the final tree (after the minibatch-std / loss edits of the end of the round)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 80 python -m pytest tests/test_networks.py tests/test_extras_gpu.py tests/test_eqlr.py tests/test_dmain_concat.py -m gpu -q -x --timeout 70 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r03_final_networks_tests.log
