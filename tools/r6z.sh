# round 6: final measurement set of the final code
mkdir -p gpurun_out/r6z
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6z/r6z_gpu_tests.txt 2>&1
tail -4 gpurun_out/r6z/r6z_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > gpurun_out/r6z/r6z_smoke.txt 2>&1
tail -2 gpurun_out/r6z/r6z_smoke.txt | cut -c1-300
sh tools/profile_round.sh r6z > gpurun_out/r6z/profile_round.log 2>&1
tail -5 gpurun_out/r6z/profile_round.log
cat gpurun_out/r6z/r6z_bench_default_wall_time.txt
python -c "
import json; d=json.load(open('gpurun_out/r6z/r6z_bench_default.json')); print(d['value'], d['ms_per_step'], d['north_star']['value'], d['small_e']['ms_per_step']); b=d['bf16_mode']; print('bf16 pair', b['f32']['ms_per_step'], b['bf16']['ms_per_step'], b['speedup_over_f32'], b['bf16']['roofline']['frac'], {k: (v['value'], v['ms_per_step']) for k, v in b['interactron'].items()}, {k: (v['value'], v['ms_per_step']) for k, v in b['interactron_bf16_fusion'].items() if isinstance(v, dict)})"
