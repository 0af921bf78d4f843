# round 6: the 16-bit activation mode end to end (adapter + native GEMM): parity test, bench line, kernel trace
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -s -k "config2_multiframe_bf16 or config2_multiframe_single_pass" > gpurun_out/r6c_b16_model_tests.txt 2>&1
tail -15 gpurun_out/r6c_b16_model_tests.txt
timeout 600 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r6c_bench_mfb_bf16.json 2> gpurun_out/r6c_bench_mfb_bf16.err
tail -3 gpurun_out/r6c_bench_mfb_bf16.err
python -c "
import json; d=json.load(open('gpurun_out/r6c_bench_mfb_bf16.json')); print(d['value'], d['ms_per_step'], d['roofline'].get('bf16_gemm'))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6c_prof -- python3 $GRAFT_REPO_ROOT/bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --step-graph off > $GRAFT_REPO_ROOT/gpurun_out/r6c_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r6c_prof -name "*kernel_stats.csv" | head -1); head -45 "$f" | cut -c1-160 > gpurun_out/r6c_kernel_stats_head.txt; cp "$f" gpurun_out/r6c_mfb_bf16_kernel_stats.csv; rm -rf gpurun_out/r6c_prof
head -40 gpurun_out/r6c_kernel_stats_head.txt
