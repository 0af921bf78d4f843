"""Meta-train step at small E: eager vs HIP-graph replay (graphs.ChunkGraphs): wall ms, host ms, loss equality in eval mode.
Usage: python tools/step_graph_probe.py [E ...]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from interactron_amd import Config, build_model
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep

Es = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
dev = torch.device("cuda")
for E in Es:
    for mode in ("false", "true"):
        cfg, _ = bench.model_cfg(300, 50, E)
        cfg["STEP_GRAPH"] = mode
        model = build_model(Config(**cfg)); load_procedural(model.fusion, "fusion."); model = model.cuda().train()
        outer = FlatOuterStep(model)
        data = bench.to_gpu(synthetic_episodes(E, height=300, width=300, tag="bench-r0"), dev)
        random.seed(0)
        for _ in range(3):
            _, losses = model(data); outer.step()
        torch.cuda.synchronize()
        hs, ws = [], []
        for _ in range(8):
            t0 = time.perf_counter()
            _, losses = model(data); outer.step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            hs.append((t1 - t0) * 1e3); ws.append((t2 - t0) * 1e3)
        st = model.__dict__.get("_chunk_graphs", {})
        print("E=%d STEP_GRAPH=%s: wall %.1f ms (min %.1f), host returns after %.1f ms; graphs: %s; loss_supervisor_ce %.5f"
              % (E, mode, sorted(ws)[4], min(ws), sorted(hs)[4], {k: type(v).__name__ for k, v in st.items()}, float(losses["loss_supervisor_ce"])), flush=True)
        del model, outer, data
        torch.cuda.empty_cache()
