mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r6t_prof2; rocprofv3 --kernel-trace --stats -d gpurun_out/r6t_prof2 -o p --output-format csv -- python3 bench.py --episodes 2 --chunk 2 --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0 > gpurun_out/r6t_e2.json 2> gpurun_out/r6t_prof2.err
cp gpurun_out/r6t_prof2/p_kernel_stats.csv gpurun_out/r6t_bench_p300_e2_kernel_stats.csv; rm -rf gpurun_out/r6t_prof2
python -c "
import json; d=json.load(open('gpurun_out/r6t_e2.json')); print(d['ms_per_step'], d['config']['step_graphs'])"
