#!/bin/sh
OUT=gpurun_out/r4g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for f in 1 0; do IX_GEMM_WP=$f timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --n800-episodes 0 > $OUT/bench_wp$f.json 2>$OUT/err_wp$f.log; python -c "
import json;d=json.load(open('$OUT/bench_wp$f.json'));r=d['roofline'];print('wp $f', round(d['value'],1), round(d['ms_per_step'],2), 'gemm ms', round(r['kernel_ms_per_step'],2), 'frac', round(r['frac'],3), 'small_e', round(d['small_e']['ms_per_step'],2))"; done
tail -3 $OUT/err_wp1.log
