# A/B inside ONE gpurun call (box-to-box variance is +-15 %): kernel modes and/or library builds
for rep in 1 2; do
for m in ${AB_MODES:-2 3}; do
for v in ${AB_LIBS:-libinteractron_hip.so}; do echo "== $v mode $m"; IX_MODE=$m IX_LIB=$v timeout 300 python tools/gemm_x6_check.py 2>&1 | grep -E "worst|BAD|tile 1128"; done
done
done
