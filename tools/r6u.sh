# round 6: MODEL.COMPUTE_DTYPE bf16_fusion (fp32-grade detector, bf16 fusion transformer): gradient fidelity of the interactron step, and the 800^2 step
mkdir -p gpurun_out
timeout 900 python -c "
import __graft_entry__ as g
for dt in ('bf16_fusion', 'bf16'):
    r = g.smoke_check(128, cfg_extra={'COMPUTE_DTYPE': dt}, f64_slack=False, norm_tol=10.0, loss_tol=1.0, cos_min=-1.0, pin_matching='always', zero_grad_noise=1e-2)
    print(dt, ': whole cosine %.5f' % r['whole_gradient_cosine'], 'worst', r['worst_cosine'], {k: round(v, 4) for k, v in list(r['loss_deviations'].items())[:5]})
" > gpurun_out/r6u_fidelity.txt 2>&1
grep -E "whole cosine|Error|error" gpurun_out/r6u_fidelity.txt | cut -c1-400
timeout 900 python bench.py --compute-dtype bf16_fusion --size 800 --episodes 8 --chunk 8 --steps 5 --warmup 2 --no-cpu-baseline --bf16-steps 0 --no-roofline > gpurun_out/r6u_bench_800_bf16_fusion.json 2> gpurun_out/r6u_bench_800_bf16_fusion.err
python -c "
import json; d=json.load(open('gpurun_out/r6u_bench_800_bf16_fusion.json')); print('interactron 800 bf16_fusion', d['value'], d['ms_per_step'], d['config']['peak_memory_GB'])"
tail -2 gpurun_out/r6u_bench_800_bf16_fusion.err
timeout 900 python bench.py --compute-dtype bf16_fusion --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 --n800-episodes 0 --stress-steps 0 --inner5-episodes 0 --no-roofline > gpurun_out/r6u_bench_300_bf16_fusion.json 2> gpurun_out/r6u_bench_300_bf16_fusion.err
python -c "
import json; d=json.load(open('gpurun_out/r6u_bench_300_bf16_fusion.json')); print('interactron 300 bf16_fusion', d['value'], d['ms_per_step'], d['small_e'] and d['small_e']['ms_per_step'])"
