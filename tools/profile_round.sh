#!/bin/sh
# One round's measurement artefacts (run on the GPU box through gpurun; results land in gpurun_out/$1/ and are copied into
# profiles/ by hand): driver-style bench line, rocprofv3 kernel stats of the same command, the two PMC HBM-traffic passes,
# SQ counters, the other reference configs / evaluation modes, the stress configuration, the 2-rank launcher smoke, power probe.
TAG=${1:-r3g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T0=$(date +%s); python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err; echo "default bench.py run: $(( $(date +%s) - T0 )) s of wall time" > $OUT/${TAG}_bench_default_wall_time.txt
rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 --small-e 0 > $OUT/${TAG}_bench_p300_e16_profiled.json 2> $OUT/prof.err
cp $OUT/prof/p_kernel_stats.csv $OUT/${TAG}_bench_p300_e16_kernel_stats.csv; rm -rf $OUT/prof
rocprofv3 --kernel-trace --stats -d $OUT/prof8 -o p --output-format csv -- python3 bench.py --size 800 --episodes 8 --chunk 8 --steps 2 --warmup 1 --no-cpu-baseline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 > $OUT/${TAG}_bench_800_e8_profiled.json 2> $OUT/prof8.err
cp $OUT/prof8/p_kernel_stats.csv $OUT/${TAG}_bench_800_e8_kernel_stats.csv; rm -rf $OUT/prof8
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 --small-e 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 --small-e 0 > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pf/p_counter_collection.csv $OUT/pw/p_counter_collection.csv $OUT/${TAG}_pmc_hbm_traffic_300.json > /dev/null; rm -rf $OUT/pf $OUT/pw
# the same two passes at the north-star shape (bench.py's n800 object reads THIS file for its `traffic`)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf8 -o p --output-format csv -- python3 bench.py --size 800 --episodes 8 --chunk 8 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw8 -o p --output-format csv -- python3 bench.py --size 800 --episodes 8 --chunk 8 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pf8/p_counter_collection.csv $OUT/pw8/p_counter_collection.csv $OUT/${TAG}_pmc_hbm_traffic_800.json > /dev/null; rm -rf $OUT/pf8 $OUT/pw8
# the 16-bit activation mode on configs[1] (multi_frame_baseline): bench line, kernel stats, the two PMC passes (bench.py's bf16_gemm.traffic reads this file)
MFB="--config multi_frame_baseline --compute-dtype bf16 --no-cpu-baseline --bf16-steps 0"
python bench.py $MFB --steps 10 --warmup 3 > $OUT/${TAG}_bench_mfb_bf16.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/profb -o p --output-format csv -- python3 bench.py $MFB --steps 6 --warmup 2 --no-roofline > /dev/null 2> $OUT/profb.err
cp $OUT/profb/p_kernel_stats.csv $OUT/${TAG}_mfb_bf16_kernel_stats.csv; rm -rf $OUT/profb
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pfb -o p --output-format csv -- python3 bench.py $MFB --steps 1 --warmup 1 --no-roofline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pwb -o p --output-format csv -- python3 bench.py $MFB --steps 1 --warmup 1 --no-roofline > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pfb/p_counter_collection.csv $OUT/pwb/p_counter_collection.csv $OUT/${TAG}_pmc_hbm_traffic_mfb_bf16.json > /dev/null; rm -rf $OUT/pfb $OUT/pwb
python bench.py --compute-dtype bf16_fusion --size 800 --episodes 8 --chunk 8 --steps 5 --warmup 2 --no-cpu-baseline --bf16-steps 0 --no-roofline > $OUT/${TAG}_bench_800_e8_bf16_fusion.json 2>/dev/null
# the per-GPU share of the reference batch on 8 GPUs (2 episodes, replayed from HIP graphs) and its neighbours
for e in 1 2 4 8; do python bench.py --episodes $e --chunk $e --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 --small-e 0 > $OUT/${TAG}_bench_p300_e$e.json 2>/dev/null; done
rocprofv3 --kernel-trace --stats -d $OUT/prof2 -o p --output-format csv -- python3 bench.py --episodes 2 --chunk 2 --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 --small-e 0 > /dev/null 2> $OUT/prof2.err
cp $OUT/prof2/p_kernel_stats.csv $OUT/${TAG}_bench_p300_e2_kernel_stats.csv; rm -rf $OUT/prof2
python tools/step_graph_probe.py 1 2 4 > $OUT/${TAG}_step_graph_probe.txt 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/ps -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 --small-e 0 > /dev/null 2>&1
python tools/pmc_sq_summary.py $OUT/ps/p_counter_collection.csv $OUT/${TAG}_pmc_sq_counters.json > /dev/null
python tools/pmc_flash_summary.py $OUT/ps/p_counter_collection.csv $OUT/${TAG}_pmc_sq_counters_flash.json > /dev/null; rm -rf $OUT/ps
for c in interactron_random multi_frame_baseline single_frame_baseline; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 > $OUT/${TAG}_bench_$c.json 2>/dev/null; done
for m in predict predict-batched interactive; do python bench.py --mode $m --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 > $OUT/${TAG}_bench_mode_$m.json 2>/dev/null; done
python bench.py --size 1600 --queries 200 --attention-dtype fp8 --episodes 1 --chunk 1 --steps 2 --warmup 1 --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 > $OUT/${TAG}_bench_stress_1600_q200_fp8.json 2>/dev/null
python bench.py --size 1600 --queries 200 --attention-dtype fp8 --mode predict --episodes 1 --steps 2 --warmup 1 --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 --no-roofline > $OUT/${TAG}_bench_stress_1600_q200_fp8_predict.json 2>/dev/null
IX_DIST_BACKEND=gloo python bench.py --gpus 2 --episodes 4 --chunk 4 --steps 2 --warmup 1 --n800-episodes 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 --no-roofline > $OUT/${TAG}_bench_2ranks_gloo_one_gpu.json 2>/dev/null
python tools/power_probe.py > $OUT/${TAG}_power_probe.txt 2>&1
ls -la $OUT
