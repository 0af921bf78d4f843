cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3g; TAG=r3g; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 > $OUT/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 > $OUT/pw.log 2>&1
ls $OUT/pf $OUT/pw; tail -3 $OUT/pf.log
python tools/pmc_summary.py $OUT/pf/p_counter_collection.csv $OUT/pw/p_counter_collection.csv $OUT/${TAG}_pmc_hbm_traffic_300.json | head -40; rm -rf $OUT/pf $OUT/pw
