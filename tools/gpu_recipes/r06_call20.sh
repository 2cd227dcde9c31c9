# round 6, call 20: whole -m gpu suite + smoke on the final tree (Python-side changes behind the record: bounds of the zero-padded epilogue tensors)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c20
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_default_noflags.json 2> $OUT/bench_default_noflags.err; echo "bench (no flags) rc=$?"; tail -1 $OUT/bench_default_noflags.json | cut -c1-400
