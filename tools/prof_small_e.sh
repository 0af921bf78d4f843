# rocprofv3 kernel stats of a small-E step: tools/prof_small_e.sh <E> <tag>
E=${1:-2}; TAG=${2:-r3a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof -o p --output-format csv -- python3 bench.py --episodes $E --chunk $E --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --n800-episodes 0 > gpurun_out/$TAG/${TAG}_bench_p300_e${E}_profiled.json 2> gpurun_out/$TAG/prof.err
cp gpurun_out/$TAG/prof/p_kernel_stats.csv gpurun_out/$TAG/${TAG}_bench_p300_e${E}_kernel_stats.csv; rm -rf gpurun_out/$TAG/prof
python bench.py --episodes $E --chunk $E --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --n800-episodes 0 > gpurun_out/$TAG/${TAG}_bench_p300_e${E}.json 2>/dev/null
cat gpurun_out/$TAG/${TAG}_bench_p300_e${E}.json | cut -c1-300
