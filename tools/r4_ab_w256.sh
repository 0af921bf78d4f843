# A/B of the 256 x 128 tiles in the whole step (same box, alternating): IX_GEMM_W256=0 / 1
mkdir -p gpurun_out/r4ab
for i in 1 2; do
  for m in 0 1; do
    IX_GEMM_W256=$m python bench.py --steps 8 --warmup 3 > gpurun_out/r4ab/bench_w$m\_$i.json 2> gpurun_out/r4ab/bench_w$m\_$i.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r4ab/bench_w${m}_$i.json").read().strip().splitlines()[-1])
print("w256=$m run $i: %.1f frames/s %.1f ms  frac %.3f  n800 %.2f (%.1f ms)  small_e %.1f ms" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["north_star"]["value"], d["north_star"]["ms_per_step"], d["small_e"]["ms_per_step"]))
PY
  done
done
