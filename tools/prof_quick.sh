cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
rocprofv3 --kernel-trace --stats -d gpurun_out/q/prof -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --n800-episodes 0 > gpurun_out/q/bench.json 2> gpurun_out/q/prof.err
cp gpurun_out/q/prof/p_kernel_stats.csv gpurun_out/q/kernel_stats.csv; rm -rf gpurun_out/q/prof
