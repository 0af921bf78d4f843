#!/bin/sh
# round 4, first GPU call: all -m gpu tests, the rccl self-test's gloo smoke, the default bench line, an E=16-only kernel trace
OUT=gpurun_out/r4a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python tools/rccl_selftest.py --gpus 2 --backend gloo --skip-bench > $OUT/selftest.log 2>&1; echo "selftest rc $?" >> $OUT/selftest.log
grep -v Warn $OUT/selftest.log | tail -8
timeout 900 python bench.py --steps 10 --warmup 3 --gemm-csv $OUT/gemm_e16.csv > $OUT/r4a_bench_default.json 2> $OUT/bench.err; echo "bench rc $?"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 > $OUT/r4a_bench_e16_only_profiled.json 2> $OUT/prof.err
cp $OUT/prof/p_kernel_stats.csv $OUT/r4a_e16_only_kernel_stats.csv; rm -rf $OUT/prof
head -c 1500 $OUT/r4a_bench_default.json
