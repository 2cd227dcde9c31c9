# round 5, GPU call 9: the driver's bench command with the captured-step companion (bench.py only changed since the final record: kernel sources and counter summary unchanged)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; print('csrc digest', c.source_digest())"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c9_bench_driver_cmd.json 2> gpurun_out/r05_c9_bench_driver_cmd.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r05_c9_bench_driver_cmd_detail.json
wc -c gpurun_out/r05_c9_bench_driver_cmd.json; cat gpurun_out/r05_c9_bench_driver_cmd.json
