"""Data parallelism end to end on the GPU: two ranks (gloo, both on GPU 0) each run ``model(shard_batch(batch))`` +
``FlatOuterStep.step()`` on their episode of a 2-episode batch; the all-reduced gradients, the update and the
PathStorage labels must equal ONE process running both episodes (reference engine/interactron_trainer.py:93-111 with
models/interactron.py:84-151 looping over the tasks of the batch)."""
import os
import random
import socket

import pytest
import torch
import torch.multiprocessing as mp

from interactron_amd.synthetic import load_procedural, synthetic_episodes
from tests.helpers import ReferenceMatching, image_key
from tests.test_parity_gpu import MODEL_CFG, to_gpu

pytestmark = pytest.mark.gpu


def _batch():
    data = synthetic_episodes(2, height=128, width=160, tag="dp")
    data["initial_image_path"] = ["dp/root", "dp/root"]   # one root image: the second episode's label depends on the first
    return data


def _build():
    from interactron_amd import Config, build_model
    m = build_model(Config(**dict(MODEL_CFG, TYPE="interactron")))
    load_procedural(m.fusion, "fusion.")
    return m.cuda().eval()


def _grads(m):
    return {k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in m.named_parameters()}


def _rank(rank, world, port, recorded, out):
    import torch.distributed as dist
    from interactron_amd.trainer import FlatOuterStep, init_distributed, shard_batch
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      IX_DIST_BACKEND="gloo")
    init_distributed()
    m = _build()
    outer = FlatOuterStep(m, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
    data = to_gpu(shard_batch(_batch(), rank, world, by_root=True))
    assert data["frames"].shape[0] == 1 and data["dp_index"] == [rank]
    # the single process draws one random first-order frame per episode, in episode order: replay this rank's draw
    random.seed(11)
    draws = [random.randint(0, 4) for _ in range(world)]
    random.randint = lambda a, b, _r=draws[rank]: _r
    with ReferenceMatching(recorded, max_flip_share=1.0, ordered=True):
        _, losses = m(data)
    outer.flat.all_reduce_grads()
    g = {k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in m.named_parameters()}
    before = outer.flat.params.detach().cpu().clone()
    total = float(outer.step(all_reduce=False))
    out[rank] = (g, total, (outer.flat.params.detach().cpu() - before).norm().item(),
                 {k: v.get_label(data["dp_actions"][1]) for k, v in m.path_storage.items()})
    dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from interactron_amd import criterion as cr
    from interactron_amd.trainer import FlatOuterStep
    # ---- one process, both episodes (episode-batched step), assignments recorded ----
    m = _build()
    outer = FlatOuterStep(m, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
    data = to_gpu(_batch())
    recorded, orig = {}, cr.HungarianMatcher.assign

    def spy(matcher, costs, targets):
        res = orig(matcher, costs, targets)
        for t, rc in zip(targets, res):
            recorded.setdefault(image_key(t), []).append(rc)
        return res

    cr.HungarianMatcher.assign = spy
    random.seed(11)
    try:
        m(data)
    finally:
        cr.HungarianMatcher.assign = orig
    ref = _grads(m)
    before = outer.flat.params.detach().cpu().clone()
    ref_total = float(outer.step())
    ref_delta = (outer.flat.params.detach().cpu() - before).norm().item()
    ref_labels = {k: v.get_label(data["actions"][1][:4].tolist()) for k, v in m.path_storage.items()}
    del m, outer
    torch.cuda.empty_cache()
    # ---- two ranks, one episode each ----
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_rank, args=(2, port, recorded, out), nprocs=2, join=True)
    (g0, t0, d0, l0), (g1, t1, d1, l1) = out[0], out[1]
    assert l0 == l1 == ref_labels                      # every rank holds the single-process trie
    assert abs(t0 - t1) <= 1e-6 * t0 and abs(d0 - d1) <= 1e-6 * max(d0, 1e-12)   # replicas stay identical
    assert abs(t0 - ref_total) <= 2e-4 * ref_total, (t0, ref_total)
    assert abs(d0 - ref_delta) <= 1e-4 * ref_delta, (d0, ref_delta)
    rel, num, den = {}, 0.0, 0.0
    for k, r in ref.items():
        a, b = g0[k], g1[k]
        assert (a is None) == (r is None), k
        if r is None:
            continue
        assert torch.equal(a, b), k                    # SUM all-reduce: bit-identical on both ranks
        n0, n1 = float(r.double().norm()), float(a.double().norm())
        d = float((a - r).double().norm())
        num, den = num + d * d, den + n0 * n0
        if max(n0, n1) < 1e-6:
            continue
        rel[k] = d / n0
    # float32 summation-order differences between the episode-batched pass and two single-episode passes (other shapes pick
    # other tiles / splits; then ~50 ReLU layers and the clipped inner step).  Every reduction is ordered, so this is a fixed
    # number, not a scatter: measured (r3, fp16x3 contraction form) total norm 7.6e-5, update 1.2e-6, whole gradient 7.4e-4,
    # median tensor 5.6e-4, worst tensor 2.5e-2 (layer3.1.conv1; bf16x6 form: 2.0e-4 / 1.1e-4 / 9.2e-3; after the cost-model retune of the
    # tile / split plan: median 2.3e-3).  The numbers move with the contraction plans (other tiles = other summation orders),
    # so the bounds leave room for that: whole 5e-3, median 5e-3, worst tensor 1e-1.  (Round 2, atomics: 3 % whole, 60 % worst.)  A missing or
    # doubled episode would put EVERY tensor off by ~50 % / 100 %.
    vals = sorted(rel.values())
    print("two ranks vs one process: total norm %.2e, update %.2e, whole gradient %.2e, median tensor %.2e, worst %.2e (%s)"
          % (abs(t0 - ref_total) / ref_total, abs(d0 - ref_delta) / ref_delta, (num / den) ** 0.5, vals[len(vals) // 2], vals[-1],
             max(rel.items(), key=lambda kv: kv[1])[0]))
    assert (num / den) ** 0.5 <= 5e-3, (num, den)
    assert vals[len(vals) // 2] <= 5e-3, vals[len(vals) // 2]
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:5]
    assert vals[-1] <= 1e-1, worst


def _rank_graph(rank, world, port, out):
    import torch.distributed as dist
    from interactron_amd import Config, build_model
    from interactron_amd.trainer import FlatOuterStep, init_distributed, shard_batch
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      IX_DIST_BACKEND="gloo")
    init_distributed()
    m = build_model(Config(**dict(MODEL_CFG, TYPE="interactron", STEP_GRAPH="true")))
    load_procedural(m.fusion, "fusion.")
    m = m.cuda().train()
    outer = FlatOuterStep(m, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
    data = to_gpu(shard_batch(_batch(), rank, world, by_root=True))
    random.seed(5)
    for _ in range(4):   # eager warm-up, capture, two replays -- each with the reward exchange between segments A and B
        m(data)
        outer.step()
    p = outer.flat.params.double()
    out[rank] = ([float(p.sum()), float((p * p).sum())], {k: v.get_label(data["dp_actions"][1]) for k, v in m.path_storage.items()},
                 sorted(type(v).__name__ for v in m.__dict__.get("_chunk_graphs", {}).values()))
    dist.destroy_process_group()


def test_two_ranks_under_graph_replay_stay_identical():
    """Data parallelism with the chunk replayed from HIP graphs (train mode, 4 steps, one episode per rank): the PathStorage
    reward exchange runs on the host between the captured segments, the flat-gradient all-reduce between the replays and the
    update -- afterwards both ranks must hold bit-identical parameters and identical tries, and both must have captured."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_rank_graph, args=(2, port, out), nprocs=2, join=True)
    (c0, l0, g0), (c1, l1, g1) = out[0], out[1]
    assert g0 == g1 == ["ChunkGraphs"], (g0, g1)
    assert c0 == c1, ("replicas drifted apart under graph replay", c0, c1)
    assert l0 == l1 and len(l0) == 1
