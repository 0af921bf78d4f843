# round 6: second-order gradients in the 16-bit / fp8 modes against the oracle: per-tensor survey (twins on / off), whole-gradient cosine
mkdir -p gpurun_out
for T in 1 0; do
IX_B16_TWINS=$T IX_SMOKE_SURVEY=1 timeout 600 python -c "
import __graft_entry__ as g
r = g.smoke_check(128, cfg_extra={'COMPUTE_DTYPE': 'bf16'}, f64_slack=False, norm_tol=1e-1, loss_tol=1.0, cos_min=-1.0, pin_matching='always', zero_grad_noise=1e-2)
print('twins $T: whole cosine', r['whole_gradient_cosine'], 'worst', r['worst_cosine'], {k: round(v, 4) for k, v in r['loss_deviations'].items()})
" > gpurun_out/r6j_survey_twins$T.txt 2>&1
grep -E "survey|twins" gpurun_out/r6j_survey_twins$T.txt | cut -c1-140 | sort -t' ' -k8 | tail -14
grep "twins" gpurun_out/r6j_survey_twins$T.txt | cut -c1-600
done
IX_SMOKE_SURVEY=1 timeout 900 python -c "
import __graft_entry__ as g
from interactron_amd import hipops
extra = dict(NUM_QUERIES=200, BLOCK_SIZE=5 * (16 * 16 + 200) + 5)
hipops.ATTENTION_DTYPE = 'fp8'
r = g.smoke_check(256, cfg_extra=extra, f64_slack=False, norm_tol=1e-1, loss_tol=1.0, cos_min=-1.0, pin_matching='always', zero_grad_noise=1e-2)
print('fp8: whole cosine', r['whole_gradient_cosine'], 'worst', r['worst_cosine'], {k: round(v, 4) for k, v in r['loss_deviations'].items()})
" > gpurun_out/r6j_survey_fp8.txt 2>&1
grep -E "survey|fp8:" gpurun_out/r6j_survey_fp8.txt | cut -c1-140 | tail -25
grep "fp8:" gpurun_out/r6j_survey_fp8.txt | cut -c1-600
