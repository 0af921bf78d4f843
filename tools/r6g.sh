# round 6: full GPU suite (every mode), the interactron step in the 16-bit mode, kernel trace of the bf16 multi_frame_baseline step
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6g_gpu_tests.txt 2>&1
tail -8 gpurun_out/r6g_gpu_tests.txt
D=$GRAFT_REPO_ROOT/gpurun_out/r6g_prof
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $D -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --step-graph off --bf16-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r6g_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r6g_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r6g_mfb_bf16_kernel_stats.csv; rm -rf gpurun_out/r6g_prof
head -30 gpurun_out/r6g_mfb_bf16_kernel_stats.csv | cut -c1-140
