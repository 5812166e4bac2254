#!/bin/bash
# Per-kernel register / scratch / occupancy report of the library (no GPU needed).
# usage: tools/resource_usage.sh [out_file]
out=$(realpath -m "${1:-/tmp/bore_resource_usage.txt}")
cd "$(dirname "$0")/../bore_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -c -fPIC -ffp-contract=off -Wno-pass-failed \
  -Rpass-analysis=kernel-resource-usage bore_all.hip -o /tmp/bore_all.o 2> /tmp/bore_ru_raw.txt
python3 - "$out" <<'PY'
import re, sys
rows, cur = [], None
for ln in open('/tmp/bore_ru_raw.txt'):
    m = re.search(r'remark: .*?Function Name: (\S+)', ln)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    m = re.search(r'remark: .*?\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)', ln)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
import subprocess
def dem(n):
    try: return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
with open(sys.argv[1], 'w') as f:
    f.write('kernel | VGPR | AGPR | SGPR | scratch B/lane | VGPR spills | SGPR spills | occupancy waves/SIMD\n')
    for r in rows:
        name = re.sub(r'^void ', '', dem(r['name'])).split('(')[0]
        f.write(f"{name} | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('SGPRs')} | {r.get('ScratchSize [bytes/lane]')} | {r.get('VGPRs Spill')} | {r.get('SGPRs Spill')} | {r.get('Occupancy [waves/SIMD]')}\n")
print(open(sys.argv[1]).read())
PY
