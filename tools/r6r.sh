# round 6: candidate final set -- full GPU suite, smoke, default bench
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6r_gpu_tests.txt 2>&1
tail -6 gpurun_out/r6r_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > gpurun_out/r6r_smoke.txt 2>&1
tail -2 gpurun_out/r6r_smoke.txt | cut -c1-300
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r6r_bench_default.json 2> gpurun_out/r6r_bench_default.err
tail -2 gpurun_out/r6r_bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r6r_bench_default.json')); print(d['value'], d['ms_per_step'], d['north_star']['value'], d['small_e']['ms_per_step']); b=d['bf16_mode']; print('bf16 pair', b['f32']['ms_per_step'], b['bf16']['ms_per_step'], b['speedup_over_f32'], b['bf16']['roofline']['frac'], {k: (v['value'], v['ms_per_step']) for k, v in b['interactron'].items()})"
