"""Per-kernel sums of one rocprofv3 SQ counter pass for the flash attention kernels.
usage: python tools/pmc_flash_summary.py <counter_collection.csv> [out.json]"""
import collections, csv, json, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(flash_\w+?_kernel<[^>]*>|attn_split_kernel<[^>]*>)", r["Kernel_Name"])
    if not m:
        continue
    k = m.group(1)
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); n[k] += 1
out = {}
for k, c in sorted(tot.items()):
    d = {"launches": n[k]}
    if c.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_per_busy_cu_cycle (of 4)"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"], 3)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_fraction"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
    if c.get("SQ_WAVE_CYCLES"):
        for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            d[w.lower() + "/wave_cycles"] = round(c[w] / c["SQ_WAVE_CYCLES"], 3)
    for w in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS"):
        if w in c:
            d[w] = c[w]
    out[k] = d
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
