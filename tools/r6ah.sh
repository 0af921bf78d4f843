mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_b16_gpu.py -q -x -k "bf16 or 16_bit or b16 or grad_views" > gpurun_out/r6ah_tests.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r6ah_tests.txt | head -12 | cut -c1-300
timeout 600 python - <<'PY'
import random, torch, bench
from interactron_amd import Config, build_model, b16
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep
cfg, _ = bench.model_cfg(300, 50, 16, "detr_multiframe", step_graph="off", compute_dtype="bf16")
m = build_model(Config(**cfg)); load_procedural(m.fusion, "fusion."); m = m.cuda().train()
outer = FlatOuterStep(m, max_norm=1.0, groups=[list(m.parameters())], lrs=[1e-5])
data = bench.to_gpu(synthetic_episodes(16, height=300, width=300, tag="casts"), torch.device("cuda", 0))
for k in range(3):
    before = dict(b16._stats)
    random.seed(k); m(data); outer.step(); torch.cuda.synchronize()
    print("step", k, {a: b16._stats[a] - before[a] for a in before})
PY
for i in 1 2; do
timeout 600 python bench.py --config multi_frame_baseline --compute-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --bf16-steps 0 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mfb bf16', round(d['ms_per_step'],2), d['config']['host_issue_ms_per_step'])"
done
