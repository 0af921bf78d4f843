#!/bin/sh
# register / scratch / LDS usage of the flash attention kernels (hipcc resource-usage remarks)
cd "$(dirname "$0")/../interactron_amd/csrc" && hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip -c flash.hip -o /tmp/flash.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    if 'error' in line: print(line.strip())
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); d={}
    for k in ('VGPRs:','AGPRs:','ScratchSize [bytes/lane]:','LDS Size [bytes/block]:'):
        if k in line and cur:
            d[k]=line.split(k)[1].split()[0]
            if k.startswith('LDS') and 'flash_' in cur: print(cur[:50], ' '.join('%s %s' % (a.split()[0], b) for a, b in d.items()))
"
