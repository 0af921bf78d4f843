# round 6: what limits the gradient fidelity of the bf16_fusion step -- arithmetic (twins off: fp32 kernels between bf16 stores; three-term attention) or storage
mkdir -p gpurun_out
for V in "IX_B16_TWINS=1 IX_B16_SINGLE_TERM=1" "IX_B16_TWINS=0 IX_B16_SINGLE_TERM=1" "IX_B16_TWINS=1 IX_B16_SINGLE_TERM=0"; do
env $V timeout 600 python -c "
import __graft_entry__ as g
for lr in (1e-3, 0.0):
    r = g.smoke_check(128, cfg_extra={'COMPUTE_DTYPE': 'bf16_fusion', 'ADAPTIVE_LR': lr}, f64_slack=False, norm_tol=10.0, loss_tol=1.0, cos_min=-1.0, pin_matching='always', zero_grad_noise=1e-2, scalar_tol=10.0)
    print('$V lr', lr, ': whole cosine %.5f' % r['whole_gradient_cosine'], 'worst', r['worst_cosine'], {k: round(v, 4) for k, v in list(r['loss_deviations'].items())[:3]})
" 2>&1 | grep "whole cosine" | cut -c1-300
done
