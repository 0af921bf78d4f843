"""CPU-only checks: the C-ABI library loads and exports every declared symbol, the host-side pieces of the hot path
(LSAP, PathStorage, config coercion, parameter plumbing, episode sharding, flat-buffer all-reduce over gloo) behave
like the reference, and the product path refuses to run without the GPU (no silent fallback)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from interactron_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODEL_CFG = dict(WEIGHTS="procedural", NUM_CLASSES=1235, SET_COST_CLASS=1.0, SET_COST_BBOX=5.0, SET_COST_GIOU=2.0,
                 NUM_LAYERS=4, NUM_HEADS=8, EMBEDDING_DIM=512, BLOCK_SIZE=2060, IMG_FEATURE_SIZE=256, OUTPUT_SIZE=512,
                 BOX_EMB_SIZE=256, EMBEDDING_PDROP=0.1, RESIDUAL_PDROP=0.1, ATTENTION_PDROP=0.1, ADAPTIVE_LR=1e-3)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    decl = _lib.parse_header()
    assert len(decl) >= 40
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.ix_version() >= 1


def test_lsap_host_matches_scipy_including_ties(lib):
    from scipy.optimize import linear_sum_assignment
    from interactron_amd import hipops
    rng = np.random.default_rng(1)
    for trial in range(600):
        nr, nc = int(rng.integers(1, 60)), int(rng.integers(1, 12))
        if trial % 3 == 0:
            nr, nc = nc, nr
        mode = trial % 4
        if mode == 0:
            c = rng.standard_normal((nr, nc)).astype(np.float32)
        elif mode == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(np.float32)
        elif mode == 2:
            c = rng.standard_normal((nr, nc)).astype(np.float32)
            if nc > 1:
                c[:, 1] = c[:, 0]
            if nr > 1:
                c[1, :] = c[0, :]
        else:
            c = np.zeros((nr, nc), np.float32)
        r, cc = hipops.lsap(torch.from_numpy(c))
        sr, sc = linear_sum_assignment(c)
        assert np.array_equal(sr, r.numpy()) and np.array_equal(sc, cc.numpy())


def test_lsap_rejects_nan(lib):
    from interactron_amd import hipops
    c = torch.zeros(3, 2)
    c[1, 1] = float("nan")
    with pytest.raises(_lib.HipLibraryError):
        hipops.lsap(c)


def test_config_coercion_rule(tmp_path):
    from interactron_amd import get_config
    p = tmp_path / "c.yaml"
    p.write_text("MODEL:\n  TYPE: \"interactron\"\n  ADAPTIVE_LR: 1e-3\n  PREDICT_ACTIONS: True\n  N: 4.0\nT:\n  LR_DECAY: Flase\n")
    cfg = get_config(str(p))
    assert cfg.MODEL.ADAPTIVE_LR == 0.001 and cfg.MODEL.PREDICT_ACTIONS == 1 and cfg.MODEL.N == 4
    assert isinstance(cfg.MODEL.N, int) and cfg.T.LR_DECAY == "Flase" and cfg.MODEL.TYPE == "interactron"
    for name in ("interactron", "interactron_random", "multi_frame_baseline", "single_frame_baseline"):
        c = get_config(os.path.join(ROOT, "configs", name + ".yaml"))
        assert c.MODEL.NUM_CLASSES == 1235


# the values the reference ships for the keys its factories switch on (utils/config_utils.py:53-113)
SHIPPED = {"single_frame_baseline": ("detr", None, "random_policy_evaluator"),   # evaluation-only in the reference
           "multi_frame_baseline": ("detr_multiframe", "direct_supervision", "random_policy_evaluator"),
           "interactron_random": ("interactron_random", "interactron_random", "random_policy_evaluator"),
           "interactron": ("interactron", "interactron", "interactive_evaluator")}


@pytest.mark.parametrize("name", sorted(SHIPPED))
def test_every_shipped_config_selects_a_valid_model_trainer_evaluator(name):
    """Every configs/*.yaml (and its *_synthetic.yaml twin) must pass the factories' arg_check, name the reference's
    types, fail loudly on a missing weight file, and the twin must differ from it in MODEL.WEIGHTS only."""
    from interactron_amd import config as cfgmod
    from interactron_amd import get_config
    from interactron_amd.episode import _load_detector_weights
    c = get_config(os.path.join(ROOT, "configs", name + ".yaml"))
    s = get_config(os.path.join(ROOT, "configs", name + "_synthetic.yaml"))
    for cfg in (c, s):
        cfgmod.arg_check(cfg.MODEL.TYPE, cfgmod.MODEL_TYPES, "model")
        trainer = cfg.TRAINER.TYPE if hasattr(cfg, "TRAINER") else None
        if trainer is not None:
            cfgmod.arg_check(trainer, ["direct_supervision", "interactron_random", "interactron"], "supervisor")
        cfgmod.arg_check(cfg.EVALUATOR.TYPE, ["random_policy_evaluator", "interactive_evaluator"], "evaluator")
        assert (cfg.MODEL.TYPE, trainer, cfg.EVALUATOR.TYPE) == SHIPPED[name]
    a, b = c.dictionarize(), s.dictionarize()
    assert b["MODEL"].pop("WEIGHTS") == "procedural" and a["MODEL"].pop("WEIGHTS").startswith("pretrained_weights/")
    assert a == b
    with pytest.raises(FileNotFoundError):
        _load_detector_weights(torch.nn.Linear(1, 1), c.MODEL)


def test_path_storage_matches_reference_script(golden):
    from interactron_amd.storage import PathStorage
    g = golden("golden_small.pt")["g14"]
    ps = PathStorage()
    for (path, rew), want in zip(g["script"], g["labels"]):
        ps.add_path(torch.tensor(path), rew)
        assert ps.get_label(path) == want


@pytest.fixture(scope="module")
def model():
    from interactron_amd import Config, build_model
    return build_model(Config(**dict(MODEL_CFG, TYPE="interactron")))


def test_module_tree_matches_reference_checkpoint_layout(golden, model):
    from interactron_amd.meta import get_parameters
    M = golden("golden_model.pt")
    named = {id(v): k for k, v in model.detector.named_parameters()}
    assert [named[id(p)] for p in get_parameters(model.detector)] == M["theta_names"]
    assert list(model.detector.state_dict().keys()) == M["detector_state_keys"]
    assert {k: tuple(v.shape) for k, v in model.fusion.state_dict().items()} == M["fusion_state_keys"]
    assert [k for k, v in model.detector.named_parameters() if v.requires_grad] == M["detector_trainable"]
    assert sum(p.numel() for p in get_parameters(model.detector)) == 38030808


def test_set_parameters_swaps_and_restores(model):
    from interactron_amd.meta import clone_parameters, detach_parameters, get_parameters, set_parameters
    theta = get_parameters(model.detector)
    repl = detach_parameters(clone_parameters(theta[:3])) + tuple(theta[3:])
    set_parameters(model.detector, repl)
    assert get_parameters(model.detector)[0] is repl[0]
    set_parameters(model.detector, theta)
    assert all(a is b for a, b in zip(get_parameters(model.detector), theta))


def test_product_path_has_no_cpu_fallback(model):
    from interactron_amd import NestedTensor
    with pytest.raises(RuntimeError):
        model.detector(NestedTensor(torch.zeros(1, 3, 64, 64), torch.zeros(1, 64, 64, dtype=torch.long)))
    from interactron_amd import hipops
    with pytest.raises(_lib.HipLibraryError):
        hipops.linear(torch.zeros(4, 8), torch.zeros(2, 8))


def test_shard_batch():
    from interactron_amd.synthetic import synthetic_episodes
    from interactron_amd.trainer import shard_batch
    data = synthetic_episodes(5, frames=2, height=8, width=8, tag="shard")
    parts = [shard_batch(data, r, 2) for r in range(2)]
    assert parts[0]["frames"].shape[0] == 3 and parts[1]["frames"].shape[0] == 2
    assert parts[1]["initial_image_path"] == ["shard/ep1", "shard/ep3"]
    assert torch.equal(parts[1]["frames"][1], data["frames"][3])
    assert torch.equal(parts[0]["boxes"][2][1], data["boxes"][4][1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from interactron_amd.trainer import FlatBuffers
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                           # replicated weights
    a, b = torch.nn.Linear(5, 3), torch.nn.Linear(3, 2)
    flat = FlatBuffers([list(a.parameters()), list(b.parameters())])
    x = torch.full((4, 5), float(rank + 1))                        # rank-dependent "episodes"
    b(a(x)).sum().backward()
    assert a.weight.grad.data_ptr() == flat.grads.data_ptr()       # grads accumulated straight into the flat buffer
    local = flat.grads.clone()
    flat.all_reduce_grads()
    out[rank] = (local, flat.grads.clone(), flat.segments)
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world_size_2_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (l0, r0, seg), (l1, r1, _) = out[0], out[1]
    assert torch.allclose(r0, l0 + l1) and torch.equal(r0, r1)   # SUM (not mean), identical on every rank
    assert seg[0][0] == 0 and seg[0][1] == seg[1][0] and seg[1][1] == r0.numel()


# ---- PathStorage under data parallelism: world_size 2 over gloo == one process ---------------------------------------
def _dp_script():
    """Three global batches of 5 / 5 / 3 episodes with repeated root images and deterministic rewards (a short last
    batch leaves rank 1 of the last chunk with fewer episodes)."""
    rng = np.random.default_rng(5)
    batches = []
    for n in (5, 5, 3):
        roots = ["root%d" % int(r) for r in rng.integers(0, 3, n)]
        actions = torch.from_numpy(rng.integers(0, 4, (n, 5)))
        rewards = [float(x) for x in rng.uniform(0.5, 3.0, n)]
        batches.append((roots, actions, rewards))
    return batches


def _dp_labels_single(chunk):
    from interactron_amd.storage import best_path_labels
    storage, labels = {}, []
    for roots, actions, rewards in _dp_script():
        for e0 in range(0, len(roots), chunk):   # the episode-batched step replays chunk by chunk, in order
            sl = slice(e0, e0 + chunk)
            labels.append(best_path_labels(storage, roots[sl], actions[sl, :4].tolist(), rewards[sl]))
    return [l for c in labels for l in c], storage


def _dp_worker(rank, world, port, chunk, out):
    import torch.distributed as dist
    from interactron_amd.episode import _Adaptive
    from interactron_amd.trainer import shard_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Host:   # the host-side half of _Adaptive.forward: chunk loop + reward exchange, no GPU work
        path_storage = {}
        _dp_chunks = staticmethod(_Adaptive._dp_chunks)
        _dp_chunk_labels = _Adaptive._dp_chunk_labels

    host, got = Host(), {}
    for roots, actions, rewards in _dp_script():
        data = {"frames": torch.zeros(len(roots), 5, 1), "actions": actions, "initial_image_path": roots}
        d = shard_batch(data, rank, world, by_root=True)
        b = d["frames"].shape[0]
        mine = d["dp_index"]
        for e0 in range(0, b, chunk):
            ep = list(range(e0, min(b, e0 + chunk)))
            lab = host._dp_chunk_labels(d, e0 // chunk, chunk, ep, [rewards[mine[t]] for t in ep])
            for t, l in zip(ep, lab):
                got[(len(got), mine[t])] = l
        for c in range((b + chunk - 1) // chunk, host._dp_chunks(d, chunk)):
            host._dp_chunk_labels(d, c, chunk, [], [])
    out[rank] = (list(got.items()), {k: v.get_label([0, 1, 2, 3]) if _has(v, [0, 1, 2, 3]) else None
                                     for k, v in host.path_storage.items()})
    dist.destroy_process_group()


def _has(store, path):
    node = store.root
    for a in path:
        if a not in node.children:
            return False
        node = node.children[a]
    return True


@pytest.mark.parametrize("chunk", [1, 2])
def test_path_storage_world_size_2_gloo_equals_single_process(chunk):
    """Ranks run episodes r::2 of every batch and exchange only the rewards; their policy labels and tries must be
    those of one process that saw the whole batch in order (reference models/interactron.py:109-115)."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_dp_worker, args=(world, port, chunk, out), nprocs=world, join=True)
    # single-process reference with the chunk size the two ranks TOGETHER cover per pass (chunk * world global episodes)
    want, storage = _dp_labels_single(chunk * world)
    per_batch, k = [], 0
    for roots, _, _ in _dp_script():
        per_batch.append(want[k:k + len(roots)])
        k += len(roots)
    seen = {0: iter(out[0][0]), 1: iter(out[1][0])}
    for bi, (roots, _, _) in enumerate(_dp_script()):
        for g in range(len(roots)):
            (_, gi), lab = next(seen[g % world])
            assert gi == g and lab == per_batch[bi][g], (bi, g, lab, per_batch[bi][g])
    assert set(out[0][1]) == set(out[1][1]) == set(storage)   # every rank holds every root's trie
    assert out[0][1] == out[1][1]


def test_reference_matching_replays_call_by_call_when_ordered(lib):
    """tests/helpers.ReferenceMatching: with two recorded assignments of ONE image that tie exactly (the 128 x 160 data-parallel
    fixture has such a pair), the default mode hands out the cheapest recorded candidate -- the same one both times -- while
    ordered=True replays the recording call by call; a recorded assignment that is not optimal is refused either way."""
    import pytest as _pt
    from interactron_amd import criterion as cr
    from tests.helpers import ReferenceMatching
    cost = torch.tensor([[1.0, 5.0], [1.0, 5.0], [9.0, 2.0]])           # queries 0 and 1 tie for target 0
    tgt = {"labels": torch.tensor([3, 4]), "boxes": torch.tensor([[0.5, 0.5, 0.2, 0.2], [0.3, 0.3, 0.1, 0.1]])}
    a = (torch.tensor([0, 2]), torch.tensor([0, 1]))
    b = (torch.tensor([1, 2]), torch.tensor([0, 1]))
    from tests.helpers import image_key
    recorded = {image_key(tgt): [b, a]}
    m = cr.HungarianMatcher()
    with ReferenceMatching(recorded, max_flip_share=1.0) as rm:
        first = m.assign([cost], [tgt])[0]
        second = m.assign([cost], [tgt])[0]
    assert torch.equal(first[0], second[0])                               # cheapest candidate, twice the same
    with ReferenceMatching(recorded, max_flip_share=1.0, ordered=True):
        first = m.assign([cost], [tgt])[0]
        second = m.assign([cost], [tgt])[0]
    assert torch.equal(first[0], b[0]) and torch.equal(second[0], a[0])   # call by call
    bad = {image_key(tgt): [(torch.tensor([0, 1]), torch.tensor([0, 1]))]}   # costs 6, the optimum 3
    with _pt.raises(AssertionError):
        with ReferenceMatching(bad, ordered=True):
            m.assign([cost], [tgt])


def test_graph_captures_hold_the_cyclic_collector_off(lib):
    """hipops.capture_begin / capture_end bracket every HIP-graph capture (graphs.ChunkGraphs, PredictGraph, the policy graph):
    dead model <-> graph cycles are collected BEFORE the capture and the collector stays off until it ends -- a CUDAGraph
    destructor running mid-capture aborts the process (round 4, gpurun_out r4r)."""
    import gc
    from interactron_amd import hipops as ops

    class Node:
        freed = []

        def __del__(self):
            Node.freed.append(gc.isenabled())

    a, b = Node(), Node()
    a.other, b.other = b, a
    del a, b
    assert gc.isenabled()
    ops.capture_begin(None)
    try:
        assert Node.freed == [True, True] and not gc.isenabled()
    finally:
        ops.capture_end()
    assert gc.isenabled()
    gc.disable()
    try:
        ops.capture_begin(None)
        ops.capture_end()
        assert not gc.isenabled()      # a caller's own setting survives
    finally:
        gc.enable()
