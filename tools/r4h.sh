#!/bin/sh
OUT=gpurun_out/r4h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 --step-graph off > $OUT/bench.json 2> $OUT/prof.err
cp $OUT/prof/p_kernel_stats.csv $OUT/kernel_stats_wp1.csv; rm -rf $OUT/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4h/kernel_stats_wp1.csv')))
for r in rows:
    if 'gemm' in r['Name'] or 'wp_' in r['Name'] or 'splitk' in r['Name']:
        print('%-80s calls/step %7.1f ms/step %8.3f avg us %8.1f'%(r['Name'][:80], int(r['Calls'])/4, float(r['TotalDurationNs'])/1e6/4, float(r['AverageNs'])/1e3))
PY
