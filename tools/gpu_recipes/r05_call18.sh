# round 5, GPU call 18: graph sets of the same step captured in one process: kernel time vs the rest, per set (tools/graph_sets_lab.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
timeout 400 python tools/graph_sets_lab.py 5 1 > gpurun_out/r05_c18_sets_stamped.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_c18_sets_stamped.log | cut -c1-700
timeout 400 python tools/graph_sets_lab.py 5 0 > gpurun_out/r05_c18_sets_clean.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_c18_sets_clean.log | cut -c1-400
