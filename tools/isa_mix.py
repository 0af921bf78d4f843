"""Instruction mix of a kernel's main loop (the backward-branch span with the most matrix instructions) from `hipcc -S` output.
usage: python tools/isa_mix.py file.s mangled_kernel_name [top]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ':')
lines = [l.strip() for l in s[i:s.index('s_endpgm', i)].split('\n')]
labels = {}
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = n
best, bm = None, -1
for n, l in enumerate(lines):
    m = re.match(r'^s_cbranch\S*\s+(\.LBB\d+_\d+)', l) or re.match(r'^s_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        nm = sum('mfma' in x for x in lines[labels[m.group(1)]:n])
        if nm > bm:
            best, bm = (labels[m.group(1)], n), nm
loop = [l for l in lines[best[0]:best[1]] if l and not l.startswith(('.', ';')) and not l.endswith(':')]
c = collections.Counter(l.split()[0] for l in loop)
grp = collections.Counter()
QUARTER = ('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u64_u32')
for k, v in c.items():
    g = ('mfma' if 'mfma' in k else 'ds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_', 'buffer_', 'scratch_'))
         else 'salu' if k.startswith('s_') else 'valu_quarter_rate' if k.startswith(QUARTER) else 'valu')
    grp[g] += v
print("loop instructions: %d  %s" % (len(loop), dict(grp)))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print("%5d %s" % (v, k))
