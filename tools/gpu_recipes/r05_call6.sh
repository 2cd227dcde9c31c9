# round 5, GPU call 6: the transposed stride-2 kernel with its stores folded into the MFMA stream (ST = 1): lab checks + timing against the burst form, the convolution tests on the
# new default, the step with SGV_CONVT_ST=1 and 0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 tools/convT_lab 5 > gpurun_out/r05_c6_convT_lab.log 2>&1; echo "lab rc=$?"
grep "check" gpurun_out/r05_c6_convT_lab.log | awk '{print $2,$3,$4,$5,$6,$7,$8,$9,"rel-L2",$(NF-9),"bad",$(NF-6)}' | sed 's/)//' | sort | uniq | head -60
grep -v check gpurun_out/r05_c6_convT_lab.log | cut -c1-150
timeout 900 python -m pytest tests/test_conv3x3_gpu.py tests/test_conv_lowp_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_conv_f16split_gpu.py tests/test_fused_conv_gpu.py tests/test_networks.py -q -m gpu --maxfail=10 -rf 2>&1 | tail -15
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0"
for st in 1 0 1 0; do
  SGV_CONVT_ST=$st timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r05_c6_bench_st$st.json 2> gpurun_out/r05_c6_bench_st$st.err
  SGV_CONVT_ST=$st python - $st <<'PY'
import json, sys
d = json.load(open('bench_detail.json'))
v = d['kernels_by_variant']
print('SGV_CONVT_ST', sys.argv[1], 'value', round(d['value'], 1), 'no_prof', round(d['value_no_prof'], 1), 'convT_ws ms', round(v['convT_ws']['ms_per_step'], 2), 'TF', round(v['convT_ws']['TFLOPs'], 1), 'packed', round(v['convT_ws_packed']['ms_per_step'], 2))
PY
done
