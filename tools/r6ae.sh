# round 6: a fixed charge for the 12-wave kernel's prologue in the tile planner (it competes with the 64 x 64 exact-fp32 kernel on small shapes)
mkdir -p gpurun_out
A="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --n800-episodes 0 --small-e 0 --bf16-steps 0 --stress-steps 0 --inner5-episodes 0"
for c in 0 8000 16000 30000 60000 0 16000; do
for e in 1 2 16; do
IX_P12_FIXED_CYCLES=$c python bench.py --episodes $e --chunk $e $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p12 fixed $c episodes $e', round(d['ms_per_step'],2))"
done; done
