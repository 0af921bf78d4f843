# round 6: the cost model's charge for a split-K reduction launch (6000 cycles since round 3; a reduction launch measures 5.5 us = 13 000)
mkdir -p gpurun_out
A="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0"
for c in 6000 3000 0 6000 3000 0; do
for e in 2 16; do
IX_SPLITK_LAUNCH_CYCLES=$c python bench.py --episodes $e --chunk $e $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cycles $c episodes $e', round(d['ms_per_step'],2))"
done; done
