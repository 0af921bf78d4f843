"""Every loop (backward-branch span) of a kernel in `hipcc -S` output with its instruction-class counts -- for kernels whose
wave roles run different loops (the 12-wave contraction kernels).  usage: python tools/isa_loops.py file.s mangled_name [min_instr]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ':')
lines = [l.strip() for l in s[i:s.index('.end_amdhsa_kernel', i) if '.end_amdhsa_kernel' in s[i:] else len(s)].split('\n')]
end = next(n for n, l in enumerate(lines) if l.startswith('.section') or l.startswith('.rodata'))
lines = lines[:end]
labels = {}
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = n
spans = []
for n, l in enumerate(lines):
    m = re.match(r'^s_cbranch\S*\s+(\.LBB\d+_\d+)', l) or re.match(r'^s_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        spans.append((labels[m.group(1)], n))
QUARTER = ('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u64_u32')
for a, b in spans:
    loop = [l for l in lines[a:b] if l and not l.startswith(('.', ';')) and not l.endswith(':')]
    if len(loop) < (int(sys.argv[3]) if len(sys.argv) > 3 else 40):
        continue
    c = collections.Counter(l.split()[0] for l in loop)
    grp = collections.Counter()
    for k, v in c.items():
        g = ('mfma' if 'mfma' in k else 'ds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_', 'buffer_', 'scratch_'))
             else 'salu' if k.startswith('s_') else 'valu_q' if k.startswith(QUARTER) else 'valu')
        grp[g] += v
    print("lines %d-%d: %d instr %s" % (a, b, len(loop), dict(grp)))
    print("    " + ", ".join("%s %d" % kv for kv in c.most_common(14)))
