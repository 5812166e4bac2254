#!/bin/bash
# register / scratch report of the fit kernels only (bore_hip.hip on its own: ~1 min)
cd "$(dirname "$0")/../bore_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -c -fPIC -ffp-contract=off --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage bore_hip.hip -o /tmp/bore_hip_only.o 2> /tmp/ru_fit_raw.txt
python3 - <<'PY'
import re
rows=[];cur=None
for ln in open('/tmp/ru_fit_raw.txt'):
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)',ln)
    if m and cur is not None: cur[m.group(1)]=int(m.group(2))
for r in rows:
    if 'fit' in r['name']:
        print(r['name'][:48], 'VGPR', r.get('VGPRs'), 'AGPR', r.get('AGPRs'), 'scratch', r.get('ScratchSize [bytes/lane]'), 'vspill', r.get('VGPRs Spill'), 'occ', r.get('Occupancy [waves/SIMD]'))
PY
grep -c " error" /tmp/ru_fit_raw.txt
