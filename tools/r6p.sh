mkdir -p gpurun_out
sh tools/gemm16_pmc.sh gpurun_out/r6p_gemm16_pmc.json 0 | tail -60
