import sys, time, torch
sys.path.insert(0, ".")
import bench
from interactron_amd import Config, build_model
from interactron_amd.synthetic import load_procedural, synthetic_episodes
cfg, _ = bench.model_cfg(300, 50, 16)
m = build_model(Config(**cfg)); load_procedural(m.fusion, "fusion."); m = m.cuda().eval()
data = bench.to_gpu(synthetic_episodes(4, height=300, width=300, tag="pp"), torch.device("cuda"))
eps = [{"frames": data["frames"][i:i+1], "masks": data["masks"][i:i+1]} for i in range(4)]
for i in range(6): m.predict(eps[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): out = m.predict(eps[i % 4])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("predict: %.2f ms per episode wall, host %.2f ms; graphs %s" % ((t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3, m.__dict__.get("_predict_graphs")))
t0 = time.perf_counter()
for i in range(20): m._graph_stamp()
print("_graph_stamp: %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
g = list(m._predict_graphs.values())[0]
t0 = time.perf_counter()
for i in range(20): g.graph.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("bare replay: %.2f ms wall, host %.2f" % ((t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3))
