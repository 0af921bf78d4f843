#!/bin/sh
OUT=gpurun_out/r4f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -s -k "single_pass or f16x3_kernel_is_fp32_grade or config2" > $OUT/pytest_sp.log 2>&1; echo "rc $?" >> $OUT/pytest_sp.log
grep "single-pass\|passed\|failed\|rror" $OUT/pytest_sp.log | head -20
for d in f32 bf16; do timeout 600 python bench.py --config multi_frame_baseline --compute-dtype $d --steps 10 --warmup 3 --no-cpu-baseline --n800-episodes 0 > $OUT/bench_mf_$d.json 2>$OUT/err_mf_$d.log; python -c "
import json;d=json.load(open('$OUT/bench_mf_$d.json'));r=d['roofline'];print('mf $d', round(d['value'],1), round(d['ms_per_step'],2), d['dtype'][:12], 'gemm ms', round(r['kernel_ms_per_step'],2), 'frac', round(r['frac'],3), 'alg TF', round(r['algorithmic_tflops'],1))"; done
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_default.json 2>$OUT/err_default.log; python -c "
import json;d=json.load(open('$OUT/bench_default.json'));r=d['roofline'];print('headline', round(d['value'],1), round(d['ms_per_step'],2), d['config']['step_graphs'], 'frac', round(r['frac'],3), 'bytes/launch', r.get('algorithmic_bytes_per_launch'), r.get('traffic_over_algorithmic'), 'n800', d['north_star'], 'small_e', d['small_e']['ms_per_step'])"
