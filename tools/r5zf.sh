#!/bin/sh
# round 5, final measurement set after the second optimisation pass (same steps as tools/r5z.sh): GPU tests, driver-style bench line, E = 16-only eager kernel trace, 800^2 trace, HBM-traffic
# PMC passes (300^2), SQ counters of the contraction kernels and of the attention kernels, other configs / modes, small-E ladder
TAG=${1:-r5zf}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; grep -v "Extension modules" $OUT/pytest.log | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
SUB="--no-cpu-baseline --n800-episodes 0 --small-e 0 --inner5-episodes 0 --stress-steps 0"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 $SUB --no-roofline --step-graph off > $OUT/${TAG}_bench_e16_only_profiled.json 2> $OUT/prof.err
cp $OUT/prof/p_kernel_stats.csv $OUT/${TAG}_e16_only_kernel_stats.csv; rm -rf $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof8 -o p --output-format csv -- python3 bench.py --size 800 --episodes 8 --chunk 8 --steps 2 --warmup 1 $SUB --no-roofline --step-graph off > $OUT/${TAG}_bench_800_e8_profiled.json 2> $OUT/prof8.err
cp $OUT/prof8/p_kernel_stats.csv $OUT/${TAG}_bench_800_e8_kernel_stats.csv; rm -rf $OUT/prof8
B="python3 bench.py --steps 1 --warmup 1 $SUB --no-roofline --step-graph off"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o p --output-format csv -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o p --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pf/p_counter_collection.csv $OUT/pw/p_counter_collection.csv $OUT/${TAG}_pmc_hbm_traffic_300.json > /dev/null; rm -rf $OUT/pf $OUT/pw
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ -d $OUT/ps -o p --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_sq_summary.py $OUT/ps/p_counter_collection.csv $OUT/${TAG}_pmc_sq_counters.json > /dev/null; rm -rf $OUT/ps
sh tools/flash_pmc.sh b800 $OUT/${TAG}_pmc_sq_counters_flash16_b800.json 1 > /dev/null 2>&1
sh tools/flash_m16_ab.sh "fusion b800" $OUT/fab > $OUT/${TAG}_flash_families_ab.txt 2>&1
for e in 1 2 4 8; do timeout 300 python bench.py --episodes $e --chunk $e --steps 10 --warmup 3 $SUB --no-roofline > $OUT/${TAG}_bench_p300_e$e.json 2>/dev/null; done
for c in interactron_random multi_frame_baseline single_frame_baseline; do timeout 300 python bench.py --config $c --steps 5 --warmup 2 $SUB > $OUT/${TAG}_bench_$c.json 2>/dev/null; done
timeout 300 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 5 --warmup 2 $SUB > $OUT/${TAG}_bench_multi_frame_baseline_bf16_single_pass.json 2>/dev/null
for m in predict predict-batched interactive; do timeout 300 python bench.py --mode $m --steps 3 --warmup 1 $SUB --no-roofline > $OUT/${TAG}_bench_mode_$m.json 2>/dev/null; done
timeout 600 python tools/rccl_selftest.py --gpus 2 --backend gloo --skip-bench > $OUT/${TAG}_rccl_selftest_gloo_2ranks_one_gpu.txt 2>/dev/null; echo "selftest rc $?"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python - $TAG <<'PY'
import json, glob, sys
tag = sys.argv[1]
for f in sorted(glob.glob('gpurun_out/%s/%s_bench_*.json' % (tag, tag))):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], round(d['value'], 1), round(d['ms_per_step'], 2), d.get('dtype', '')[:10])
    except Exception as e: print(f, 'ERR', e)
d = json.load(open('gpurun_out/%s/%s_bench_default.json' % (tag, tag)))
print('north_star', d['north_star']); print('small_e', d['small_e']['ms_per_step'], d['small_e']['strong_scaling_projection'])
r = d['roofline']; print('frac', r['frac'], 'kernel ms', r['kernel_ms_per_step'], 'attn ms', r['attention_kernels']['kernel_ms_per_step'], 'bytes', r['algorithmic_bytes_per_launch'], r['traffic_over_algorithmic'])
print('n800 attn', d['n800']['roofline']['attention_kernels']['kernel_ms_per_step'], 'n800 gemm', d['n800']['roofline']['kernel_ms_per_step'], d['n800']['roofline']['frac'])
print('inner5', d['inner5']); print('stress', d['stress']); print('cpu', d['cpu_baseline'])
PY
