#!/bin/bash
# Register / scratch report of selected kernels for an experiment build (no GPU needed).
# usage: tools/ru_quick.sh <kernel-name-regex> [extra hipcc flags...]
pat=$1; shift
cd "$(dirname "$0")/../bore_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -c -fPIC -ffp-contract=off --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" bore_all.hip -o /tmp/ru_quick_$$.o 2> /tmp/ru_quick_$$.txt
python3 - /tmp/ru_quick_$$.txt "$pat" <<'PY'
import re, sys
rows, cur = [], None
for ln in open(sys.argv[1]):
    if 'error' in ln: print(ln.rstrip())
    m = re.search(r'remark: .*?Function Name: (\S+)', ln)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    m = re.search(r'remark: .*?\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)', ln)
    if m and cur is not None: cur[m.group(1)] = int(m.group(2))
for r in rows:
    if re.search(sys.argv[2], r['name']):
        print(f"{r['name'][:60]:60s} VGPR {r.get('VGPRs')} AGPR {r.get('AGPRs')} scratch {r.get('ScratchSize [bytes/lane]')} vspill {r.get('VGPRs Spill')} sspill {r.get('SGPRs Spill')} occ {r.get('Occupancy [waves/SIMD]')}")
PY
rm -f /tmp/ru_quick_$$.o /tmp/ru_quick_$$.txt
