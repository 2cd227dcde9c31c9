# round 5, GPU call 25: captured and eager blocks alternating in one process, with the card's clock and power per block (tools/eager_vs_graph_state_lab.py), twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
for r in 1 2; do timeout 300 python tools/eager_vs_graph_state_lab.py > gpurun_out/r05_c25_eager_vs_graph_$r.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_c25_eager_vs_graph_$r.log | tail -18 | cut -c1-300; done
