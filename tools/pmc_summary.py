"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/<tag>_pmc_hbm_traffic.json: HBM bytes per launch of every contraction and flash attention kernel.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys

def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(gemm_f32_\w+?kernel|gemm_wp_kernel|gemm16_\w*?kernel|flash\w*?_kernel\w*)", r["Kernel_Name"])
        if not m:
            continue
        tot[m.group(1)] += float(r["Counter_Value"]) * 1024.0   # both counters are in KiB
        n[m.group(1)] += 1
    return tot, n

fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 "
                  "--no-cpu-baseline --no-roofline",
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16-B/lane streaming loads, MI355X_MICROARCH.md HBM "
                     "section); both counters in KiB",
       "kernels": {}}
for k in sorted(fetch):
    rd = 2.0 * fetch[k] / max(1, nf[k])
    wr = write.get(k, 0.0) / max(1, nw.get(k, 0))
    out["kernels"][k] = {"launches": nf[k], "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                         "hbm_bytes_per_launch": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
