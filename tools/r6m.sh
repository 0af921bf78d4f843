# round 6: what one bf16 GEMM launch costs without one of its parts (make diag16) on the step's Linear shapes
mkdir -p gpurun_out
python tools/gemm16_bench.py --json gpurun_out/r6m_g16_full.json > gpurun_out/r6m_g16_full.txt 2>&1
for d in 1 2 3; do
IX_LIB_PATH=$PWD/interactron_amd/lib/libix_g16_diag$d.so python tools/gemm16_bench.py --json gpurun_out/r6m_g16_diag$d.json > gpurun_out/r6m_g16_diag$d.txt 2>&1
done
python - <<'PY'
import json
runs = {k: json.load(open('gpurun_out/r6m_g16_%s.json' % k)) for k in ('full', 'diag1', 'diag2', 'diag3')}
print('%-4s %6s %5s %6s | %8s %8s %8s %8s' % ('kind', 'M', 'N', 'K', 'full', 'no-mfma', 'no-dma', 'no-store'))
for i, r in enumerate(runs['full']['shapes']):
    print('%-4s %6d %5d %6d | %8.1f %8.1f %8.1f %8.1f' % (r['kind'], r['M'], r['N'], r['K'], r['us'], runs['diag1']['shapes'][i]['us'], runs['diag2']['shapes'][i]['us'], runs['diag3']['shapes'][i]['us']))
for k, v in runs.items():
    print(k, {a: round(b['ms_per_step'], 2) for a, b in v['summary'].items()})
PY
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_b800_gpu.py -m gpu -q -k "fp8" > gpurun_out/r6m_fp8_tests.txt 2>&1
tail -4 gpurun_out/r6m_fp8_tests.txt
