import os, sys, random
sys.path.insert(0, os.getcwd())
import torch, bench
from interactron_amd import Config, build_model
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep
cfg, _ = bench.model_cfg(300, 50, 16)
model = build_model(Config(**cfg)); load_procedural(model.fusion, "fusion."); model = model.cuda().train()
outer = FlatOuterStep(model)
data = bench.to_gpu(synthetic_episodes(16, height=300, width=300, tag="bench-r0"), torch.device("cuda"))
random.seed(0)
for _ in range(2):
    model(data); outer.step()
torch.cuda.synchronize()
print("peak allocated %.1f GB, reserved %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30, torch.cuda.max_memory_reserved() / 2**30))
