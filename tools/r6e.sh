# round 6: 16-bit mode with convolution gathers
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_b16_gpu.py -q > gpurun_out/r6e_b16_tests.txt 2>&1
tail -12 gpurun_out/r6e_b16_tests.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -k "config2_multiframe_bf16" > gpurun_out/r6e_b16_model_tests.txt 2>&1
grep -v "Warn\|warn" gpurun_out/r6e_b16_model_tests.txt | tail -12
timeout 600 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r6e_bench_mfb_bf16.json 2> gpurun_out/r6e_bench_mfb_bf16.err
python -c "
import json; d=json.load(open('gpurun_out/r6e_bench_mfb_bf16.json')); print(d['value'], d['ms_per_step'], d['roofline'].get('bf16_gemm'))"
D=$GRAFT_REPO_ROOT/gpurun_out/r6e_prof
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $D -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --step-graph off > $GRAFT_REPO_ROOT/gpurun_out/r6e_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r6e_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r6e_mfb_bf16_kernel_stats.csv; rm -rf gpurun_out/r6e_prof
head -40 gpurun_out/r6e_mfb_bf16_kernel_stats.csv | cut -c1-150
