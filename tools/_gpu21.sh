cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export SGV_LIB=$GRAFT_REPO_ROOT/stylegan-v_amd/csrc/libsgv_hip.so
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "upfirdn2d" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/t21.log
for st in 8 16 32; do echo "== strip $st"; SGV_LANES_STRIP=$st timeout 60 ./tools/ufd_lab 32 2>&1 | grep -E "libsgv"; done | tee gpurun_out/ufd_lab_lds.log
timeout 200 python tools/ops_bench.py --frames 32 --reps 20 --only upfirdn2d 2>&1 | grep -v amdgpu.ids | head -12 | tee gpurun_out/ops_bench_lds.log
cd /tmp
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lds -- $GRAFT_REPO_ROOT/tools/ufd_lab 32 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pmc_lds/*/*counter_collection.csv')[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'lanes' in r['Kernel_Name']]
print('LDS hand-off: FETCH_SIZE mean KB', sum(v)/len(v), ' x2 / algorithmic read =', 2*sum(v)/len(v)/528392)
PY
