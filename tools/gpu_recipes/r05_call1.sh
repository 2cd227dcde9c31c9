# round 5, GPU call 1: the driver's bench command with the compact contract line + the consumer-mapping lab of the transposed stride-2 kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 tools/convT_lab 5 > gpurun_out/r05_c1_convT_lab.log 2>&1; echo "lab rc=$?"
grep -c "check" gpurun_out/r05_c1_convT_lab.log; grep "check" gpurun_out/r05_c1_convT_lab.log | awk '{print $2,$3,$4,$5,$6,$7,$8,$9, $(NF-11), $(NF-10), $(NF-9), "bad", $(NF-6)}' | sort | uniq -c | sort -rn | head -50
grep -v check gpurun_out/r05_c1_convT_lab.log | cut -c1-120
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c1_bench.json 2> gpurun_out/r05_c1_bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r05_c1_bench.json; wc -c gpurun_out/r05_c1_bench.json
