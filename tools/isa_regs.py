"""Registers / scratch / LDS of every kernel in `hipcc -S` output (the .amdgpu_metadata block).  usage: python tools/isa_regs.py file.s [substr]"""
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: +\d+", s, flags=re.S):
    blk = m.group(0)
    name = re.search(r"\.name: +(\S+)", blk).group(1)
    if sub not in name:
        continue
    get = lambda k: int(re.search(r"\.%s: +(\d+)" % k, blk).group(1))
    print("%-52s vgpr %3d  agpr %3d  spill %3d  scratch %4d B  lds %6d B" % (name, get("vgpr_count"), get("agpr_count"), get("vgpr_spill_count"),
                                                                        get("private_segment_fixed_size"), get("group_segment_fixed_size")))
