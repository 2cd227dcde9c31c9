# round 5, GPU call 19: graph sets again, 3 per process (4 sets = 254 GiB of private pools: call 18 ran out of memory at the fifth); does a slow set show up with the
# stamps in (kernel time by family of a slow vs a fast set), how often is the first set the slow one, and what does the runtime's packet capture do
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
for r in 1 2; do timeout 300 python tools/graph_sets_lab.py 3 1 > gpurun_out/r05_c19_stamped_$r.log 2>&1; grep "^set\|^again\|^memory\|Error" gpurun_out/r05_c19_stamped_$r.log | cut -c1-420; done
for r in 1 2; do timeout 300 python tools/graph_sets_lab.py 3 0 > gpurun_out/r05_c19_clean_$r.log 2>&1; grep "^set\|^again\|^memory\|Error" gpurun_out/r05_c19_clean_$r.log | cut -c1-300; done
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python tools/graph_sets_lab.py 3 0 > gpurun_out/r05_c19_clean_nopkt.log 2>&1; grep "^set\|^again\|^memory\|Error" gpurun_out/r05_c19_clean_nopkt.log | cut -c1-300
