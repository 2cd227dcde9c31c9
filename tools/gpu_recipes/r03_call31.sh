# round 3, GPU call 31: what bounds the FIR tile kernel on 16-bit tensors -- lab builds without the multiply-adds (1), without the clamp-undo selects (2), without both (3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cp stylegan-v_amd/csrc/libsgv_hip.so /tmp/base.so
B="python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --steps 8 --warmup 3"
for lib in base abl1 abl2 abl3; do
  if [ $lib = base ]; then cp /tmp/base.so stylegan-v_amd/csrc/libsgv_hip.so; else cp tools/exp/libsgv_$lib.so stylegan-v_amd/csrc/libsgv_hip.so; fi
  for mode in "--lowp bf16" ""; do
  timeout 200 $B $mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kv=d['kernels_by_variant']
print('$lib', '$mode', round(d['value'],1), {k: (round(v['ms_per_step'],2), round(v['GBps'])) for k,v in kv.items() if k.startswith('ufd_tile')})"
  done
done | tee gpurun_out/r03_ufd_tile_valu_ablation.log
cp /tmp/base.so stylegan-v_amd/csrc/libsgv_hip.so
