#!/bin/bash
# tools/kres.sh build.log  -> one line per physics kernel: VGPRs, spills, scratch, occupancy, LDS (from -Rpass-analysis=kernel-resource-usage output)
python3 - "$1" <<'PY'
import re, sys
cur = None; d = {}
for line in open(sys.argv[1]):
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = m.group(1); d[cur] = {}; continue
    m = re.search(r'remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)', line)
    if m and cur: d[cur][m.group(1)] = int(m.group(2))
for k, v in d.items():
    if re.search(r'k_physics|k_bd_physics|k_bd_settle', k):
        print('%-58s' % k[:58], ' '.join('%s=%s' % (a.split(' ')[0], b) for a, b in v.items()))
PY
