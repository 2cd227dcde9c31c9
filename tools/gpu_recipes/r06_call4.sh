# round 6, call 4: steady-state round-robin protocol (24 warm + 24 timed launches per variant and round, 5 rounds, median): the product tile kernel's features switched
# off one at a time (lab builds), V6 with the XCD order as the control inside every process
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c4
mkdir -p $OUT
export TMPDIR=/tmp
for a in "96 257 1" "96 256 2" "32 257 1" "32 256 2"; do
  echo "== $a  full table (LAB_OFF=0)" >> $OUT/ufd_lab6.log
  timeout 300 tools/ufd_lab6_off0 $a >> $OUT/ufd_lab6.log 2>&1
  for v in 2 4 8 16 32; do
    echo "== $a  LAB_OFF=$v" >> $OUT/ufd_lab6.log
    UFD_ONLY="V7 product" UFD_ONLY2="NTload0 xcd1" timeout 300 tools/ufd_lab6_off$v $a 2>&1 | grep -v "^FIR" >> $OUT/ufd_lab6.log
  done
done
cat $OUT/ufd_lab6.log | cut -c1-200
