# round 6: 16-bit mode after the bias-gradient fusion; the interactron configurations in the mode
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_b16_gpu.py -q > gpurun_out/r6h_b16_tests.txt 2>&1
tail -5 gpurun_out/r6h_b16_tests.txt
timeout 1200 python -m pytest tests -m gpu -q -s -k "interactron_step_in_the_16_bit or fp8_attention_against or config2_multiframe_bf16" > gpurun_out/r6h_model_tests.txt 2>&1
grep -v "Warn\|warn" gpurun_out/r6h_model_tests.txt | grep -E "passed|failed|16-bit mode|Assertion|bf16" | tail -12
timeout 600 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 > gpurun_out/r6h_bench_mfb_bf16.json 2> gpurun_out/r6h_bench_mfb_bf16.err
python -c "
import json; d=json.load(open('gpurun_out/r6h_bench_mfb_bf16.json')); print('mfb bf16', d['value'], d['ms_per_step'], d['roofline']['bf16_gemm']['kernel_ms_per_step'], d['roofline']['bf16_gemm']['frac'])"
timeout 900 python bench.py --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 --n800-episodes 0 --stress-steps 0 --inner5-episodes 0 > gpurun_out/r6h_bench_interactron_bf16.json 2> gpurun_out/r6h_bench_interactron_bf16.err
python -c "
import json; d=json.load(open('gpurun_out/r6h_bench_interactron_bf16.json')); print('interactron 300 bf16', d['value'], d['ms_per_step'], d['config']['step_graphs'])"
tail -3 gpurun_out/r6h_bench_interactron_bf16.err
timeout 900 python bench.py --compute-dtype bf16 --size 800 --episodes 8 --chunk 8 --steps 5 --warmup 2 --no-cpu-baseline --bf16-steps 0 > gpurun_out/r6h_bench_interactron_bf16_800.json 2> gpurun_out/r6h_bench_interactron_bf16_800.err
python -c "
import json; d=json.load(open('gpurun_out/r6h_bench_interactron_bf16_800.json')); print('interactron 800 bf16', d['value'], d['ms_per_step'], d['config']['peak_memory_GB'])"
tail -3 gpurun_out/r6h_bench_interactron_bf16_800.err
