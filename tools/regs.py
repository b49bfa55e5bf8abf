#!/usr/bin/env python
"""Register / spill summary from `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr saved to a file).
    python tools/regs.py remarks.txt [name-substring ...]"""
import re, sys
cur = None; rows = {}
for line in open(sys.argv[1]):
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r':\s+(VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if len(sys.argv) > 2 and not any(f in k for f in sys.argv[2:]):
        continue
    print(f"{k[:70]:70s} V={v.get('VGPRs')} A={v.get('AGPRs')} spill={v.get('VGPRs Spill')} scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')}")
