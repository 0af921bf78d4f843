# round 6, second GPU call: the bf16 GEMM's op tests and shape bench; the re-bounded second-order tests
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_b16_gpu.py -q -x > gpurun_out/r6b_b16_tests.txt 2>&1
tail -5 gpurun_out/r6b_b16_tests.txt
timeout 600 python tools/gemm16_bench.py --json gpurun_out/r6b_gemm16_bench.json > gpurun_out/r6b_gemm16_bench.txt 2>&1
tail -30 gpurun_out/r6b_gemm16_bench.txt
timeout 1200 python -m pytest tests -m gpu -q -k "g13 or config3 or inner_steps or fp8_attention_against" > gpurun_out/r6b_rebounded.txt 2>&1
tail -5 gpurun_out/r6b_rebounded.txt
