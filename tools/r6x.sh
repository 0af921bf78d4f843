mkdir -p gpurun_out
python tools/gemm_census.py --size 300 --episodes 2 --top 60 --out gpurun_out/r6x_gemm_census_300_e2.json > gpurun_out/r6x_gemm_census_300_e2.txt 2>&1
tail -70 gpurun_out/r6x_gemm_census_300_e2.txt | cut -c1-200
