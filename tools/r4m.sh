#!/bin/sh
# PMC evidence for round 4: HBM traffic per launch (300^2, 800^2; separate FETCH / WRITE passes), SQ counters of the step's
# contraction kernels, and SQ counters of the two contraction kernels on ONE identical shape (tools/wp_one.py)
TAG=r4m; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 --step-graph off"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o p --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pf/p_counter_collection.csv $OUT/pw/p_counter_collection.csv $OUT/${TAG}_pmc_hbm_traffic_300.json > /dev/null; rm -rf $OUT/pf $OUT/pw
B8="python3 bench.py --size 800 --episodes 8 --chunk 8 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 --step-graph off"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf8 -o p --output-format csv -- $B8 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw8 -o p --output-format csv -- $B8 > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pf8/p_counter_collection.csv $OUT/pw8/p_counter_collection.csv $OUT/${TAG}_pmc_hbm_traffic_800.json > /dev/null; rm -rf $OUT/pf8 $OUT/pw8
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQ -d $OUT/ps -o p --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_sq_summary.py $OUT/ps/p_counter_collection.csv $OUT/${TAG}_pmc_sq_counters.json > /dev/null; rm -rf $OUT/ps
rocprofv3 --kernel-trace --pmc $SQ -d $OUT/p1 -o p --output-format csv -- python3 tools/wp_one.py > $OUT/wp_one.log 2>&1
python tools/pmc_sq_summary.py $OUT/p1/p_counter_collection.csv $OUT/${TAG}_pmc_sq_counters_one_shape_1805x2048x256x16.json; rm -rf $OUT/p1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d $OUT/p2 -o p --output-format csv -- python3 tools/wp_one.py > $OUT/wp_one2.log 2>&1
python - <<'PY'
import csv, collections, json, re
try:
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open('gpurun_out/r4m/p2/p_counter_collection.csv')):
        m = re.search(r"(gemm_f32_\w+?kernel|gemm_wp_kernel)", r["Kernel_Name"])
        if m: tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
    json.dump({k: dict(v) for k, v in tot.items()}, open('gpurun_out/r4m/r4m_pmc_sq_lds_counters_one_shape.json', 'w'), indent=1)
    print(json.dumps({k: dict(v) for k, v in tot.items()}, indent=1))
except Exception as e:
    print("second SQ pass failed:", e)
PY
rm -rf $OUT/p2; tail -2 $OUT/wp_one.log; ls $OUT
