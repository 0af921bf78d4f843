# round 6: candidate final set -- full GPU suite, smoke, default bench
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6i_gpu_tests.txt 2>&1
tail -6 gpurun_out/r6i_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > gpurun_out/r6i_smoke.txt 2>&1
tail -2 gpurun_out/r6i_smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6i_bench_default.json 2> gpurun_out/r6i_bench_default.err
tail -2 gpurun_out/r6i_bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r6i_bench_default.json')); print(d['value'], d['ms_per_step'], d['north_star']['value'], d['small_e']['ms_per_step']); print(json.dumps(d['bf16_mode'], indent=1)[:3000])"
