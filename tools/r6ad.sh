# round 6: the 16-bit mode's tests with the bf16 GEMM forced onto its other forms (256-tile form everywhere; stores from the MFMA layout; two LDS stages)
mkdir -p gpurun_out
for V in "IX_GEMM16_BIG=2" "IX_GEMM16_CST=0" "IX_GEMM16_STAGES=2 IX_GEMM16_BIG=0"; do
env $V timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_b16_gpu.py -q -k "bf16 or 16_bit or b16" > gpurun_out/r6ad_tests.txt 2>&1
echo "$V: $(grep -E 'passed|failed' gpurun_out/r6ad_tests.txt | tail -1)"
done
