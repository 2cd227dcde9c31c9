# round 4, GPU call 10: aten-level profile of two PLAIN iterations (no regularisation phases): who issues the torch launches of the regular step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SGV_TORCH_PROFILE=gpurun_out/r04_c10_torch_profile_plain.txt SGV_TORCH_PROFILE_FROM=1 timeout 240 python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --steps 2 --warmup 3 --no-prof > /dev/null 2> gpurun_out/r04_c10.err; echo "rc=$?"
python - <<'PY'
import re
txt = open('gpurun_out/r04_c10_torch_profile_plain.txt').read()
part = txt[txt.index('by call count'):]
rows = []
for line in part.splitlines()[1:]:
    m = re.match(r'(.{90})\s+(\d+)\s+([\d.]+)\s+([\d.]+)', line)
    if m:
        rows.append((float(m.group(4)), int(m.group(2)), float(m.group(3)), m.group(1).strip()))
rows.sort(reverse=True)
print('top by device ms (2 plain steps):')
for dev, calls, cpu, name in rows[:60]:
    print('%9.3f ms %6d calls  cpu %8.2f  %s' % (dev, calls, cpu, name[:100]))
PY
