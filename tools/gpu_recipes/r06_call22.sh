# round 6, call 22: grouped-affine backward with a column shared three ways; bias_act's partial buffer from the zero arena: the touched suites
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c22
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_fc_gpu.py tests/test_ops_gpu.py tests/test_networks.py tests/test_extras_gpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest.log; tail -3 $OUT/pytest.log
