# round 6, call 13: same-box A/B of two builds of the library (SGV_LIB_PATH): base = commit "2x tile kernels", new = hoisted scale/bias loads + reciprocal act-gradient
# math + one bound atomic per workgroup + mode 2 on the tile kernel
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c13
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for v in base new; do
  SGV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lab/libsgv_$v.so timeout 600 python tools/fir_bench.py --frames 96 --amax 1 --rounds 3 > $OUT/fir_n96_amax_${v}_$rep.log 2>> $OUT/err.log
done
done
SGV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lab/libsgv_new.so SGV_UFD_TILE_EPI2=0 timeout 600 python tools/fir_bench.py --frames 96 --amax 1 --rounds 3 --only mode2 > $OUT/fir_n96_amax_new_mode2lanes.log 2>> $OUT/err.log
python - <<'PY'
import re, glob
def load(f):
    d={}
    for l in open(f):
        m=re.match(r'(.+?)\s+([\d.]+) us', l)
        if m and not l.startswith('#'): d[m.group(1).strip()]=float(m.group(2))
    return d
b1,n1,b2,n2=[load(f'gpurun_out/r06_c13/fir_n96_amax_{v}_{r}.log') for v,r in (('base',1),('new',1),('base',2),('new',2))]
ml=load('gpurun_out/r06_c13/fir_n96_amax_new_mode2lanes.log')
print(f"{'case':52s} base1   new1   base2   new2   new/base")
for k in b1:
    print(f"{k:52s} {b1[k]:7.1f} {n1[k]:7.1f} {b2[k]:7.1f} {n2[k]:7.1f}  {(n1[k]+n2[k])/(b1[k]+b2[k]):.3f}" + (f"   (new, mode 2 on the lane-exchange kernel: {ml[k]:.1f})" if k in ml else ''))
PY
tail -2 $OUT/err.log
