import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from interactron_amd import Config, build_model
from interactron_amd.synthetic import load_procedural, synthetic_episodes
cfg, _ = bench.model_cfg(300, 50, 16)
model = build_model(Config(**cfg)); load_procedural(model.fusion, "fusion."); model = model.cuda().eval()
data = bench.to_gpu(synthetic_episodes(2, height=300, width=300, tag="g"), torch.device("cuda"))
def ep(i, s): return {"frames": data["frames"][i:i+1, :s], "masks": data["masks"][i:i+1, :s]}
for use in (True, False):
    model.config.POLICY_GRAPH = use
    for s in (1, 2, 3, 4): model.get_next_action(ep(0, s))
    torch.cuda.synchronize(); t = time.perf_counter()
    acts = []
    for r in range(5):
        for s in (1, 2, 3, 4): acts.append(model.get_next_action(ep(r % 2, s)))
    torch.cuda.synchronize()
    print("graph" if use else "eager", "%.2f ms per call" % ((time.perf_counter() - t) / 20 * 1e3), acts[:8], "graphs:", None if model._graphs is None else len(model._graphs))
