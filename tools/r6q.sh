# round 6: bf16 mode after the LDS store pass, the flat bf16 parameter shadow and the 256-tile form where it wins
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_b16_gpu.py -q > gpurun_out/r6q_b16_tests.txt 2>&1
tail -3 gpurun_out/r6q_b16_tests.txt
python tools/gemm16_bench.py --json gpurun_out/r6q_gemm16_bench.json > gpurun_out/r6q_gemm16_bench.txt 2>&1
tail -22 gpurun_out/r6q_gemm16_bench.txt | grep -A30 "^{" | grep -E "ms_per_step|tflops"
timeout 1200 python -m pytest tests -m gpu -q -s -k "config2_multiframe_bf16 or interactron_step_in_the_16_bit or fp8" > gpurun_out/r6q_model_tests.txt 2>&1
grep -v "Warn\|warn" gpurun_out/r6q_model_tests.txt | grep -E "passed|failed|Assertion|16-bit mode" | tail -8 | cut -c1-400
timeout 600 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 > gpurun_out/r6q_bench_mfb_bf16.json 2> gpurun_out/r6q_bench_mfb_bf16.err
python -c "
import json; d=json.load(open('gpurun_out/r6q_bench_mfb_bf16.json')); print('mfb bf16', d['value'], d['ms_per_step'], d['roofline']['bf16_gemm']['kernel_ms_per_step'], d['roofline']['bf16_gemm']['frac'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r6q_prof; rocprofv3 --kernel-trace --stats -d gpurun_out/r6q_prof -o p --output-format csv -- python3 bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --bf16-steps 0 --no-roofline --step-graph off > gpurun_out/r6q_prof.log 2>&1
cp $(find gpurun_out/r6q_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r6q_mfb_bf16_kernel_stats.csv; rm -rf gpurun_out/r6q_prof
head -30 gpurun_out/r6q_mfb_bf16_kernel_stats.csv | cut -c1-150
