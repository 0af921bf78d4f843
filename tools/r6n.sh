# round 6: bf16 GEMM epilogue storing whole row pieces through LDS
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_b16_gpu.py -q > gpurun_out/r6n_b16_tests.txt 2>&1
tail -5 gpurun_out/r6n_b16_tests.txt
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_b800_gpu.py -m gpu -q -k "fp8" > gpurun_out/r6n_fp8_tests.txt 2>&1
tail -3 gpurun_out/r6n_fp8_tests.txt
python tools/gemm16_bench.py --json gpurun_out/r6n_gemm16_bench.json > gpurun_out/r6n_gemm16_bench.txt 2>&1
tail -22 gpurun_out/r6n_gemm16_bench.txt
IX_GEMM16_CST=0 python tools/gemm16_bench.py --json gpurun_out/r6n_gemm16_bench_cst0.json > gpurun_out/r6n_gemm16_bench_cst0.txt 2>&1
tail -22 gpurun_out/r6n_gemm16_bench_cst0.txt | grep -A30 "^{"
timeout 1200 python -m pytest tests -m gpu -q -s -k "config2_multiframe_bf16" > gpurun_out/r6n_model_tests.txt 2>&1
grep -v "Warn\|warn" gpurun_out/r6n_model_tests.txt | grep -E "passed|failed|Assertion" | tail -5
timeout 600 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 > gpurun_out/r6n_bench_mfb_bf16.json 2> gpurun_out/r6n_bench_mfb_bf16.err
python -c "
import json; d=json.load(open('gpurun_out/r6n_bench_mfb_bf16.json')); print('mfb bf16', d['value'], d['ms_per_step'], d['roofline']['bf16_gemm']['kernel_ms_per_step'], d['roofline']['bf16_gemm']['frac'])"
