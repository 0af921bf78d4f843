# round 6: A/B of the headline step, the tree before the hipops package split (_ab_old, commit 76cd534) against this one, same box
mkdir -p gpurun_out
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --bf16-steps 0 --n800-episodes 0 --stress-steps 0 --inner5-episodes 0 --small-e 0 --no-roofline"
for i in 1 2; do
(cd _ab_old && timeout 600 python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'], d['config']['host_issue_ms_per_step'])")
timeout 600 python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'], d['config']['host_issue_ms_per_step'])"
done
