rm -f gpurun_out/survey.txt
for i in 1 2; do IX_TEST_RECORD=gpurun_out/survey.txt timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -s 2>&1 | grep -E "passed|failed|ReferenceMatching|Error|assert" | tail -20; done
sort -k1,1 -k2,2gr gpurun_out/survey.txt | awk '{k=$1" "$2; if (!seen[$1" "$2" "$3]++) print}' | sort -k2,2gr | head -40
