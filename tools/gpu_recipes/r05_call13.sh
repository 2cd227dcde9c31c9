# round 5, GPU call 13: the event nodes through torch's capture (tools/graph_event_repro.py), return codes printed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SGV_DEBUG_EVENTS=1 timeout 120 python tools/graph_event_repro.py > gpurun_out/r05_c13_repro.log 2>&1; echo rc=$?; tail -25 gpurun_out/r05_c13_repro.log | cut -c1-300
