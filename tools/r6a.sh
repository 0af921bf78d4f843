# round 6, first GPU call: tolerance survey at rel = 1e-3 (+ float64 slack), the new parity tests, the full -m gpu suite, the default bench
mkdir -p gpurun_out
rm -f gpurun_out/r6a_survey.txt
IX_TEST_RECORD=gpurun_out/r6a_survey.txt IX_TEST_RECORD_ALL=1 timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -k "g13 or config3" > gpurun_out/r6a_survey_pytest.txt 2>&1
IX_SMOKE_SURVEY=1 timeout 600 python -c "
import __graft_entry__ as g
g.smoke_check(128)
g.smoke_check(128, inner_steps=2, chunk=16)
" > gpurun_out/r6a_smoke_survey.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -s -k "g5 or g6 or fp8" > gpurun_out/r6a_new_tests.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6a_gpu_tests.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6a_bench_default.json 2> gpurun_out/r6a_bench_default.err
tail -3 gpurun_out/r6a_gpu_tests.txt
