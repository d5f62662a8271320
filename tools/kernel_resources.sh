#!/bin/bash
# Register / scratch / LDS summary of every kernel of one .hip source (no GPU needed):
#   tools/kernel_resources.sh dca_amd/csrc/dcahip_heads.hip [filter]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$(dirname "$0")/../include" -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage ${HIPCC_EXTRA} 2>&1 | python3 -c "
import sys, re
cur = None; rows = {}
for line in sys.stdin:
    if 'error' in line: print(line, end='')
    m = re.search(r'remark: +Function Name: (\S+)', line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    if not re.search(r'$filt', name): continue
    print('%-90s VGPR %3d  AGPR %3d  scratch %4d B  spillV %3d  LDS %6d  SGPR %3d' % (name[:90], v.get('VGPRs', -1), v.get('AGPRs', 0), v.get('ScratchSize', -1), v.get('VGPRs Spill', -1), v.get('LDS Size', -1), v.get('TotalSGPRs', -1)))
"
