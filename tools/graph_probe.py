"""Round-3 probe: what does a small-E meta-train step cost to ISSUE, and what would a HIP-graph replay of it cost?
For each E: (a) eager step: wall ms, host ms until model(data)+outer.step() return; (b) the sync-free core of the step
(expand, detector fwd, fusion fwd, inner gradient with create_graph, clipped SGD, detector fwd, a stand-in total, the
second-order backward) issued eagerly vs replayed from one captured HIP graph.
Usage: python tools/graph_probe.py [E ...]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from interactron_amd import Config, build_model, hipops as ops
from interactron_amd.detector import NestedTensor
from interactron_amd.meta import set_parameters, sgd_step
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep

Es = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]
dev = torch.device("cuda")
for E in Es:
    cfg, _ = bench.model_cfg(300, 50, E)
    model = build_model(Config(**cfg)); load_procedural(model.fusion, "fusion."); model = model.cuda().train()
    outer = FlatOuterStep(model)
    data = bench.to_gpu(synthetic_episodes(E, height=300, width=300, tag="bench-r0"), dev)
    random.seed(0)
    for _ in range(2):
        model(data); outer.step()
    torch.cuda.synchronize()
    hs, ws = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        model(data); outer.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append((t1 - t0) * 1e3); ws.append((t2 - t0) * 1e3)
    print("E=%d eager step: wall %.1f ms (min %.1f), host returns after %.1f ms" % (E, sorted(ws)[2], min(ws), sorted(hs)[2]), flush=True)

    s = 5
    frames = data["frames"].reshape(E * s, 3, 300, 300)
    masks = data["masks"].reshape(E * s, 300, 300)
    theta = model._real_parameters()
    model._theta = theta
    targets2 = model._second_order_targets()
    w1 = torch.randn(E * s, 50, 1236, device=dev)
    w2 = torch.randn(E * s, 50, 4, device=dev)

    def core():
        dtheta = [t.requires_grad_(True) for t in ops.ExpandEpisodes.apply(E, *[p.detach() for p in theta])]
        set_parameters(model.detector, dtheta)
        nt = NestedTensor(frames, masks)
        nt.stem = model.detector.backbone[0].body.frozen_stem(frames)
        pre = model.detector(nt)
        pre = {k: v.reshape((E, s) + tuple(v.shape[1:])) for k, v in pre.items()}
        fusion_out = model.fusion(pre)
        loss_map = fusion_out["loss"].reshape(E, -1)
        learned = torch.stack([ops.l2_norm(loss_map[i]) for i in range(E)]).sum()
        grads = model._inner_grad(learned, dtheta, True)
        set_parameters(model.detector, sgd_step(dtheta, grads, 1e-3))
        post = model.detector(nt)
        total = ops.Dot.apply(post["pred_logits"], w1) + ops.Dot.apply(post["pred_boxes"], w2) \
            + ops.Dot.apply(fusion_out["actions"].reshape(-1), w2.reshape(-1)[:E * 16])
        with ops.skip_param_grads(frozenset(id(t) for t in dtheta)):
            torch.autograd.backward(total, inputs=targets2)
        set_parameters(model.detector, theta)
        return total

    try:
        for _ in range(2):
            core()
        torch.cuda.synchronize()
        hs, ws = [], []
        lib = ops._L()
        for _ in range(5):
            t0 = time.perf_counter(); core(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            hs.append((t1 - t0) * 1e3); ws.append((t2 - t0) * 1e3)
        print("E=%d core eager: wall %.1f ms (min %.1f), host %.1f ms" % (E, sorted(ws)[2], min(ws), sorted(hs)[2]), flush=True)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            core()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        t0 = time.perf_counter()
        with torch.cuda.graph(g):
            out = core()
        torch.cuda.synchronize()
        print("E=%d capture took %.1f ms; pool %.2f GB" % (E, (time.perf_counter() - t0) * 1e3, torch.cuda.memory_reserved() / 1e9), flush=True)
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        hs, ws = [], []
        for _ in range(5):
            t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            hs.append((t1 - t0) * 1e3); ws.append((t2 - t0) * 1e3)
        print("E=%d core graph replay: wall %.1f ms (min %.1f), host %.1f ms; total=%s" % (E, sorted(ws)[2], min(ws), sorted(hs)[2], float(out)), flush=True)
        del g
    except Exception as e:   # noqa
        import traceback; traceback.print_exc()
        torch.cuda.synchronize()
    finally:
        set_parameters(model.detector, theta)
    del model, outer, data
    torch.cuda.empty_cache()
