"""Which tensors does autograd sum with aten::add / add_ in one meta-train step (shapes, counts)?"""
import os, random, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from interactron_amd import Config, build_model
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep
cfg, _ = bench.model_cfg(300, 50, 16, step_graph="off")   # (eager: a replayed graph shows no ATen ops)
model = build_model(Config(**cfg)); load_procedural(model.fusion, "fusion."); model = model.cuda().train()
outer = FlatOuterStep(model)
data = bench.to_gpu(synthetic_episodes(16, height=300, width=300, tag="bench-r0"), torch.device("cuda"))
random.seed(0)
for _ in range(2):
    model(data); outer.step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    model(data); outer.step(); torch.cuda.synchronize()
agg = collections.Counter(); byt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_") and e.input_shapes:
        shp = tuple(e.input_shapes[0])
        n = 1
        for d in shp: n *= d
        agg[(e.name, shp)] += 1
        byt[(e.name, shp)] += n * 4 * 3
tot = sum(byt.values())
print("total add traffic %.1f GB" % (tot / 1e9))
for k, v in byt.most_common(25):
    print("%-10s %-28s x%4d  %.2f GB" % (k[0], k[1], agg[k], v / 1e9))
