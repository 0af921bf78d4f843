# round 6: two-stage bf16 GEMM A/B, op tests in both forms, the mode's model test
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_b16_gpu.py -q > gpurun_out/r6f_b16_tests.txt 2>&1
tail -6 gpurun_out/r6f_b16_tests.txt
for S in 1 2 1 2; do timeout 300 python tools/gemm16_bench.py --stages $S --json gpurun_out/r6f_gemm16_bench_s$S.json > gpurun_out/r6f_gemm16_bench_s$S.txt 2>&1; python -c "
import json; d=json.load(open('gpurun_out/r6f_gemm16_bench_s$S.json'))['summary']; print('stages $S', {k: (round(v['ms_per_step'],2), round(v['tflops'])) for k,v in d.items()})"; done
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -k "config2_multiframe_bf16" > gpurun_out/r6f_b16_model_tests.txt 2>&1
grep -v "Warn\|warn" gpurun_out/r6f_b16_model_tests.txt | tail -8
for S in 1 2; do IX_GEMM16_STAGES=$S timeout 600 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r6f_bench_mfb_bf16_s$S.json 2> gpurun_out/r6f_bench_mfb_bf16_s$S.err; python -c "
import json; d=json.load(open('gpurun_out/r6f_bench_mfb_bf16_s$S.json')); print('stages $S', d['value'], d['ms_per_step'], d['roofline']['bf16_gemm']['kernel_ms_per_step'], d['roofline']['bf16_gemm']['frac'])"; done
