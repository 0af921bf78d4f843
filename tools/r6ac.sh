mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -k "fusion_transformer_in_the_16_bit" > gpurun_out/r6ac_tests.txt 2>&1
grep -E "passed|failed|whole-gradient|Assertion|smoke ok" gpurun_out/r6ac_tests.txt | cut -c1-400
