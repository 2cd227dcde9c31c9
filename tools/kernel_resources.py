#!/usr/bin/env python3
"""Per-kernel register / spill / LDS table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).

    python tools/kernel_resources.py stylegan-v_amd/csrc/conv3x3.hip [name-filter]

The producer / consumer kernels are tuned to sit exactly at 256 registers without spills; run this after every edit of theirs."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
res = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
if res.returncode != 0:
    sys.exit(res.stderr[-3000:])
cur, rows = None, []
for line in res.stderr.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        full = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = dict(name=full[:full.rfind('(')] if full.endswith(')') else full)
        rows.append(cur)
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\d+) \[-Rpass', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print('%-100s %5s %5s %5s %6s %7s %7s %4s' % ('kernel', 'SGPR', 'VGPR', 'AGPR', 'spill', 'scratch', 'LDS', 'occ'))
for r in rows:
    if flt in r['name']:
        print('%-100s %5d %5d %5d %6d %7d %7d %4d' % (r['name'][-100:], r.get('TotalSGPRs', -1), r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('VGPRs Spill', -1), r.get('ScratchSize', -1),
                                                   r.get('LDS Size', -1), r.get('Occupancy', -1)))
