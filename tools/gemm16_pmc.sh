# SQ counters of the bf16 GEMM kernels over tools/gemm16_bench.py.  usage: sh tools/gemm16_pmc.sh <out.json> [IX_GEMM16_BIG]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/gemm16_pmc.json}; export IX_GEMM16_BIG=${2:-0}
D=gpurun_out/g16pmc_tmp; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $D/a -o p --output-format csv -- python3 tools/gemm16_bench.py --iters 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $D/b -o p --output-format csv -- python3 tools/gemm16_bench.py --iters 2 > /dev/null 2>&1
python3 - $D $OUT <<'PY'
import collections, csv, glob, json, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for f in glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm16\w*_kernel<[^>]*>)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1); tot[k][r["Counter_Name"] + ("" if "/a/" in f else "#b")] += float(r["Counter_Value"])
        if "/a/" in f and (r["Dispatch_Id"], k) not in seen: seen.add((r["Dispatch_Id"], k)); n[k] += 1
out = {}
for k, c in sorted(tot.items()):
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    d = {"launches": n[k], "mfma_busy_per_busy_cu_cycle_of_4": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["SQ_BUSY_CU_CYCLES"], 1), 3),
         "lds_conflict_fraction": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), 4),
         "lds_active_per_busy_cu_cycle": round(c["SQ_LDS_IDX_ACTIVE"] / max(c["SQ_BUSY_CU_CYCLES"], 1), 3),
         "wait_any": round(c["SQ_WAIT_ANY"] / wc, 3), "wait_inst_any": round(c["SQ_WAIT_INST_ANY"] / wc, 3), "active_inst_any": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3)}
    wb = c["SQ_WAVE_CYCLES#b"] or 1.0
    for w in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"):
        d[w.lower() + "_per_mfma"] = round(c[w + "#b"] / max(c["SQ_INSTS_MFMA#b"], 1), 3)
    d["wave_quadcycles_per_mfma"] = round(wb / max(c["SQ_INSTS_MFMA#b"], 1), 2)
    d["wait_inst_lds"] = round(c["SQ_WAIT_INST_LDS#b"] / wb, 3); d["active_inst_valu"] = round(c["SQ_ACTIVE_INST_VALU#b"] / wb, 3)
    out[k] = d
json.dump(out, open(sys.argv[2], "w"), indent=1); print(json.dumps(out, indent=1))
PY
rm -rf $D
