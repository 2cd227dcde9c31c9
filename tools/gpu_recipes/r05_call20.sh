# round 5, GPU call 20: the two speeds of the same graph (140 / 149 ms) against the engine clock, socket power and temperature sampled while it replays
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -40 > gpurun_out/r05_c20_hwmon_files.txt
(rocm-smi --showperflevel --showmaxpower --showclocks --showpower --showtemp 2>&1 | head -60) > gpurun_out/r05_c20_rocm_smi_idle.txt
for r in 1 2 3; do timeout 300 python tools/graph_sets_lab.py 3 0 > gpurun_out/r05_c20_clean_$r.log 2>&1; grep "^set\|^again\|Error" gpurun_out/r05_c20_clean_$r.log | cut -c1-700; done
timeout 300 python tools/graph_sets_lab.py 2 1 > gpurun_out/r05_c20_stamped.log 2>&1; grep "^set\|^again\|Error" gpurun_out/r05_c20_stamped.log | cut -c1-900
cat gpurun_out/r05_c20_rocm_smi_idle.txt | head -40
