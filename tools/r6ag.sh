mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "without_persistent_grad_views or config2 or config1 or single_frame" > gpurun_out/r6ag_tests.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r6ag_tests.txt | head -12 | cut -c1-300
for S in 0 1 0 1; do
IX_STEAL_GRADS=$S timeout 600 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steal $S mfb bf16', round(d['ms_per_step'],2), d['config']['host_issue_ms_per_step'])"
done
for S in 0 1; do
IX_STEAL_GRADS=$S timeout 600 python bench.py --config multi_frame_baseline --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steal $S mfb f32', round(d['ms_per_step'],2))"
IX_STEAL_GRADS=$S timeout 600 python bench.py --config single_frame_baseline --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steal $S sfb f32', round(d['ms_per_step'],2))"
done
