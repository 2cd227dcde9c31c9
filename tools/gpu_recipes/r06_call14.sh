# round 6, call 14: whole -m gpu suite + smoke + the driver's bench command on the tree of commit "bench.py: the captured step is the headline at every N ..."
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c14
mkdir -p $OUT/tables
export TMPDIR=/tmp
SGV_ERROR_TABLE_DIR=$OUT/tables timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
cp bench_detail.json $OUT/bench_driver_cmd_detail.json
tail -1 $OUT/bench_driver_cmd.json | cut -c1-2800
