#!/bin/sh
OUT=gpurun_out/r4d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 300 python tools/add_shapes.py > $OUT/add_shapes_fanout.txt 2>/dev/null
IX_FANOUT=0 timeout 300 python tools/add_shapes.py > $OUT/add_shapes_plain.txt 2>/dev/null
head -30 $OUT/add_shapes_fanout.txt; head -8 $OUT/add_shapes_plain.txt
for f in 1 0; do IX_FANOUT=$f timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --n800-episodes 0 > $OUT/bench_fanout$f.json 2>/dev/null; python -c "
import json;d=json.load(open('$OUT/bench_fanout$f.json'));print('fanout $f', d['value'], d['ms_per_step'], d['small_e']['ms_per_step'])"; done
