cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export SGV_LIB=$GRAFT_REPO_ROOT/stylegan-v_amd/csrc/libsgv_hip.so
SGV_LANES_WPB=8 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "upfirdn2d" 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2 3; do for w in 4 8; do echo -n "wpb=$w 257->256: "; SGV_LANES_WPB=$w timeout 60 ./tools/ufd_lab 32 2>&1 | grep -E "libsgv" | cut -c36-; done; done | tee gpurun_out/ufd_lab_wpb_ab.log
for rep in 1 2; do for w in 4 8; do echo -n "wpb=$w N=96: "; SGV_LANES_WPB=$w timeout 60 ./tools/ufd_lab 96 2>&1 | grep -E "libsgv" | cut -c36-; done; done | tee -a gpurun_out/ufd_lab_wpb_ab.log
