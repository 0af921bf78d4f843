mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r6w_prof; rocprofv3 --kernel-trace --stats -d gpurun_out/r6w_prof -o p --output-format csv -- python3 bench.py --compute-dtype bf16_fusion --size 800 --episodes 8 --chunk 8 --steps 2 --warmup 1 --no-cpu-baseline --bf16-steps 0 --no-roofline --n800-episodes 0 > gpurun_out/r6w_800.json 2> gpurun_out/r6w_prof.err
cp gpurun_out/r6w_prof/p_kernel_stats.csv gpurun_out/r6w_800_e8_bf16_fusion_kernel_stats.csv; rm -rf gpurun_out/r6w_prof
