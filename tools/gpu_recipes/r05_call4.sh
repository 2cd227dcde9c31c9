# round 5, GPU call 4: whole -m gpu suite after the fixes of call 3 (edge strips of the transposed fp16 form use fp16-rounded weights; dispatch expectations), the driver's bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SGV_ERROR_TABLE_DIR=gpurun_out timeout 1500 python -m pytest tests/ -q -m gpu --maxfail=40 -rf -s 2>&1 > gpurun_out/r05_c4_pytest_full.log; grep "fp16 golden" gpurun_out/r05_c4_pytest_full.log | cut -c1-200; grep -v "^\[\|^\.\|^$" gpurun_out/r05_c4_pytest_full.log | tail -50
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c4_bench.json 2> gpurun_out/r05_c4_bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r05_c4_bench_detail.json
tail -c 2600 gpurun_out/r05_c4_bench.json
