# round 6: kernel stats and PMC traffic of configs[1] in the 16-bit mode on the final tree
OUT=gpurun_out/r6zy; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MFB="--config multi_frame_baseline --compute-dtype bf16 --no-cpu-baseline --bf16-steps 0"
python bench.py $MFB --steps 10 --warmup 3 > $OUT/r6zy_bench_mfb_bf16.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/profb -o p --output-format csv -- python3 bench.py $MFB --steps 6 --warmup 2 --no-roofline > /dev/null 2> $OUT/profb.err
cp $OUT/profb/p_kernel_stats.csv $OUT/r6zy_mfb_bf16_kernel_stats.csv; rm -rf $OUT/profb
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pfb -o p --output-format csv -- python3 bench.py $MFB --steps 1 --warmup 1 --no-roofline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pwb -o p --output-format csv -- python3 bench.py $MFB --steps 1 --warmup 1 --no-roofline > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pfb/p_counter_collection.csv $OUT/pwb/p_counter_collection.csv $OUT/r6zy_pmc_hbm_traffic_mfb_bf16.json > /dev/null; rm -rf $OUT/pfb $OUT/pwb
python -c "
import json,csv; d=json.load(open('$OUT/r6zy_bench_mfb_bf16.json')); print(d['value'], d['ms_per_step'], d['roofline']['bf16_gemm']['frac'], d['roofline']['bf16_gemm']['traffic'], d['config']['host_issue_ms_per_step'])
rows=list(csv.DictReader(open('$OUT/r6zy_mfb_bf16_kernel_stats.csv'))); print(sum(float(r['TotalDurationNs']) for r in rows)/1e6/8, sum(int(r['Calls']) for r in rows)/8)"
