# round 6: where does the direction error of the interactron step in the 16-bit mode come from?  (no adaptation / fp32-grade reference run)
mkdir -p gpurun_out
for LR in 0.0 0.001; do
timeout 600 python -c "
import __graft_entry__ as g
for dt in ('f32', 'bf16'):
    r = g.smoke_check(128, cfg_extra={'COMPUTE_DTYPE': dt, 'ADAPTIVE_LR': $LR}, f64_slack=False, norm_tol=10.0, loss_tol=1.0, cos_min=-1.0, pin_matching='always', zero_grad_noise=1e-2)
    print('LR $LR', dt, ': whole cosine %.5f' % r['whole_gradient_cosine'], 'worst', r['worst_cosine'], {k: round(v, 4) for k, v in list(r['loss_deviations'].items())[:4]})
" > gpurun_out/r6k_lr$LR.txt 2>&1
grep "whole cosine" gpurun_out/r6k_lr$LR.txt | cut -c1-300
done
