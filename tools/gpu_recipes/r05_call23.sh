# round 5, GPU call 23: the other clock domains (memory, fabric, SoC) of the card in the slow and in the fast state: four short processes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
ls /sys/class/drm/card*/device/ | grep -i "pp_\|power_dpm\|gpu_busy\|mem_busy" | sort -u | head -30 > gpurun_out/r05_c23_sysfs_files.txt
for r in 1 2 3 4; do timeout 200 python tools/graph_state_lab.py 80 > gpurun_out/r05_c23_state_$r.log 2>&1; grep -v amdgpu.ids gpurun_out/r05_c23_state_$r.log | cut -c1-700; done
cat gpurun_out/r05_c23_sysfs_files.txt
