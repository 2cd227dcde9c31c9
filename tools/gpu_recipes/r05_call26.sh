# round 5, GPU call 26: stride-1 convolution back to back, workgroups started in phase (0) or staggered (1, 2, 4, 8): time per launch and the card's clock / power
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
for k in 0 1 2 4 8 0; do SGV_CONV_STAGGER=$k timeout 200 python tools/stagger_lab.py 2>&1 | grep "^stagger" | cut -c1-300; done | tee gpurun_out/r05_c26_stagger.log
