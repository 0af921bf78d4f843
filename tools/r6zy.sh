# round 6: the last validation of the final tree (full GPU suite, smoke, the driver-style bench line)
mkdir -p gpurun_out/r6zy
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6zy/r6zy_gpu_tests.txt 2>&1
tail -3 gpurun_out/r6zy/r6zy_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > gpurun_out/r6zy/r6zy_smoke.txt 2>&1
tail -1 gpurun_out/r6zy/r6zy_smoke.txt
T0=$(date +%s); timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r6zy/r6zy_bench_default.json 2> gpurun_out/r6zy/bench.err; echo "wall $(( $(date +%s) - T0 )) s"
python -c "
import json; d=json.load(open('gpurun_out/r6zy/r6zy_bench_default.json')); print(d['value'], d['ms_per_step'], d['north_star']['value'], d['small_e']['ms_per_step'], d['roofline']['frac']); b=d['bf16_mode']; print('bf16 pair', b['f32']['ms_per_step'], b['bf16']['ms_per_step'], b['speedup_over_f32'], b['bf16']['roofline']['frac'], b['bf16']['roofline']['traffic'], {k: (v['value'], v['ms_per_step']) for k, v in b['interactron'].items()}, {k: (v['value'], v['ms_per_step']) for k, v in b['interactron_bf16_fusion'].items() if isinstance(v, dict)})"
