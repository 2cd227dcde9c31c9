# round 5, GPU call 22: the slow / fast state of the captured step over a minute of replays (tools/graph_state_lab.py), twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
for r in 1 2; do timeout 400 python tools/graph_state_lab.py > gpurun_out/r05_c22_state_$r.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_c22_state_$r.log | cut -c1-1200; done
