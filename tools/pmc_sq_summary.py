"""Per-kernel sums of one rocprofv3 SQ counter pass (MFMA busy, waits, LDS conflicts) for the contraction kernels.
usage: python tools/pmc_sq_summary.py <counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(gemm_f32_\w+?kernel|gemm_wp_kernel)", r["Kernel_Name"])
    if not m:
        continue
    k = m.group(1)
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); n[k] += 1
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                  "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline",
       "units": "MFMA_BUSY in cycles (32 per 32x32x16 bf16 MFMA); WAVE/WAIT/ACTIVE in quad-cycles summed over waves", "kernels": {}}
for k, c in tot.items():
    d = dict(c); d["launches"] = n[k]
    if c.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_over_busy_cu_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAVE_CYCLES"):
        for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            d[w.lower() + "_over_wave_cycles"] = c[w] / c["SQ_WAVE_CYCLES"]
    out["kernels"][k] = d
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
