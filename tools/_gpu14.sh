cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export SGV_LIB=$GRAFT_REPO_ROOT/stylegan-v_amd/csrc/libsgv_hip.so
cd /tmp
for shape in "32 257 1" "32 256 2"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_$tag -- $GRAFT_REPO_ROOT/tools/ufd_lab $shape > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_$tag.log 2>&1
  timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write_$tag -- $GRAFT_REPO_ROOT/tools/ufd_lab $shape > $GRAFT_REPO_ROOT/gpurun_out/pmc_write_$tag.log 2>&1
done
ls $GRAFT_REPO_ROOT/gpurun_out | grep pmc
