# round 6: the 256 x 256-tile form of the bf16 GEMM
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_b16_gpu.py -q -x > gpurun_out/r6o_b16_tests.txt 2>&1
tail -8 gpurun_out/r6o_b16_tests.txt
python tools/gemm16_bench.py --json gpurun_out/r6o_gemm16_bench.json > gpurun_out/r6o_gemm16_bench.txt 2>&1
tail -22 gpurun_out/r6o_gemm16_bench.txt | grep -A30 "^{" | grep -E "ms_per_step|tflops"
IX_GEMM16_BIG=0 python tools/gemm16_bench.py --json gpurun_out/r6o_gemm16_bench_big0.json > gpurun_out/r6o_gemm16_bench_big0.txt 2>&1
tail -22 gpurun_out/r6o_gemm16_bench_big0.txt | grep -A30 "^{" | grep -E "ms_per_step|tflops"
