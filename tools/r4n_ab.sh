cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
for v in new old new old; do
  if [ $v = old ]; then export IX_LIB_PATH=$GRAFT_REPO_ROOT/interactron_amd/lib/libix_flash_old.so; else unset IX_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/r4n/p_$v -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 --step-graph off > /dev/null 2>&1
  grep "attn_split_kernel" gpurun_out/r4n/p_$v/p_kernel_stats.csv | sed "s/^/$v /" | cut -c1-160
  rm -rf gpurun_out/r4n/p_$v
done
