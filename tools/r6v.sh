mkdir -p gpurun_out
IX_SMOKE_SURVEY=1 timeout 900 python -c "
import __graft_entry__ as g
r = g.smoke_check(128, cfg_extra={'COMPUTE_DTYPE': 'bf16_fusion'}, f64_slack=False, norm_tol=5e-2, loss_tol=1.0, cos_min=-1.0, pin_matching='always', zero_grad_noise=1e-2)
print('whole cosine %.5f' % r['whole_gradient_cosine'], r['worst_cosine'])
" > gpurun_out/r6v_survey.txt 2>&1
grep "survey:" gpurun_out/r6v_survey.txt | sort -k7 -g | tail -12 | cut -c1-160
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -k "fusion_transformer_in_the_16_bit or 16_bit_mode_against" > gpurun_out/r6v_tests.txt 2>&1
grep -E "passed|failed|whole-gradient|Assertion" gpurun_out/r6v_tests.txt | cut -c1-300
