mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r6af_gpu_tests.txt 2>&1
grep -A30 "slowest" gpurun_out/r6af_gpu_tests.txt | head -32
tail -2 gpurun_out/r6af_gpu_tests.txt
