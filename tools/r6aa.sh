mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "chunk_graph_replay_equals" > gpurun_out/r6aa_tests.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r6aa_tests.txt | head -20 | cut -c1-300
