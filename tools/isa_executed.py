"""Instructions a wave EXECUTES per trip of a kernel's main loop in `hipcc -S` output: the loop body without the blocks behind
`s_cbranch_vccz` (the rare range-check paths of csrc/flash16.hip), by class.  usage: python tools/isa_executed.py file.s [name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else "flash16"
for name in sorted(set(re.findall(r"^(_Z\w+):", s, flags=re.M))):
    if sub not in name:
        continue
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
    if best is None:
        continue
    out, n = [], best[0]
    while n < best[1]:
        l = lines[n]
        m = re.match(r'^s_cbranch_vccz\s+(\.LBB\d+_\d+)', l)
        if m and n < labels[m.group(1)] <= best[1]:
            out.append(l)
            n = labels[m.group(1)]
            continue
        if l and not l.startswith(('.', ';')) and not l.endswith(':'):
            out.append(l)
        n += 1
    c = collections.Counter()
    for l in out:
        k = l.split()[0]
        c['mfma' if 'mfma' in k else 'ds_tr' if k.startswith('ds_read_b64_tr') else 'ds' if k.startswith('ds_') else
          'vmem' if k.startswith(('global_', 'buffer_', 'scratch_')) else 'waitcnt' if k == 's_waitcnt' else 'nop' if k == 's_nop'
          else 'salu' if k.startswith('s_') else 'trans' if k.startswith(('v_exp', 'v_mad_u64', 'v_mul_lo', 'v_mul_hi')) else 'valu'] += 1
    print("%-52s executed %4d (%.2f per matrix instruction)  %s" % (name[:52], len(out), len(out) / max(c['mfma'], 1), dict(c)))
