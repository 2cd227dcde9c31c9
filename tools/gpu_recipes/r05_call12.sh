# round 5, GPU call 12: events inside captured graphs on this stack (tools/graph_event_lab.hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 tools/graph_event_lab > gpurun_out/r05_c12_graph_event_lab.log 2>&1; echo rc=$?; cat gpurun_out/r05_c12_graph_event_lab.log
