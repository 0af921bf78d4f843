"""RCCL self-test for the data-parallel path -- for the first box with more than one GPU.

    python tools/rccl_selftest.py --gpus N [--backend nccl|gloo] [--skip-bench]

The launcher (this process) NEVER touches a GPU: it starts N ranks as child processes (`torch.distributed.run`, rendezvous
on 127.0.0.1), waits, then starts `bench.py --gpus N --global-batch 16` as another child, and exits with the first
non-zero code.  Each rank (`--worker`) checks, over RCCL (backend "nccl" on ROCm) unless `--backend gloo`:

  1. the step's ONE data-path collective as the trainer issues it: `FlatBuffers.all_reduce_grads()` on a flat fp32 buffer of
     the real size (58.4 M floats = 234 MB): SUM semantics (rank-dependent fill -> closed-form expectation), bit-identical
     result on every rank (MIN == MAX of a checksum), and its time / bus bandwidth over 5 runs;
  2. the PathStorage reward exchange of `episode._Adaptive._dp_chunk_labels` (a few floats per chunk through pinned staging):
     labels and tries equal to a single process replaying the global batch;
  3. a HIP-graph capture (`capture_error_mode="thread_local"`) of a small kernel sequence WHILE the RCCL watchdog thread is
     alive, an all-reduce between two replays, replays bit-equal to eager;
  4. two meta-train steps of the real model on `shard_batch` shards (2 episodes per rank at 128 x 160, graphs on):
     parameters bit-identical across ranks afterwards (`_check_replicas`' checksum).

Prints one JSON line per rank 0 check; exit code 0 = all passed.  With one visible GPU and `--backend gloo` the ranks share
the GPU (a plumbing smoke test, what `gpurun` boxes can run)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launcher(args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IX_DIST_BACKEND=args.backend, PYTHONPATH=ROOT)
    port = str(29000 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.abspath(__file__), "--worker", "--gpus", str(args.gpus), "--backend", args.backend]
    rc = subprocess.call(cmd, env=env, cwd=ROOT)
    if rc != 0 or args.skip_bench:
        return rc
    bench = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
             "127.0.0.1", "--master-port", str(int(port) + 1), os.path.join(ROOT, "bench.py"), "--gpus", str(args.gpus),
             "--global-batch", "16", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--n800-episodes", "0"]
    return subprocess.call(bench, env=env, cwd=ROOT)


def report(rank, name, ok, **kw):
    if rank == 0:
        print(json.dumps(dict(check=name, ok=bool(ok), **kw)), flush=True)
    return bool(ok)


def worker(args):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from interactron_amd.trainer import FlatBuffers, init_distributed, shard_batch
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    rank, _, world = init_distributed(args.backend)
    dev = torch.device("cuda", local)
    ok = True

    def fence():
        dist.barrier(device_ids=[local]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    # 1. the flat gradient all-reduce at its real size
    n = 58_400_000
    p = torch.nn.Parameter(torch.zeros(n, device=dev))
    flat = FlatBuffers([[p]])
    idx = torch.arange(n, device=dev, dtype=torch.float32)
    flat.grads.copy_((idx % 1024) * (rank + 1))          # exact in fp32 for world <= 64
    flat.all_reduce_grads()
    want = (idx % 1024) * (world * (world + 1) // 2)
    exact = bool(torch.equal(flat.grads[:n], want))
    chk = flat.grads.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    fence()
    t0 = time.perf_counter()
    for _ in range(5):
        flat.all_reduce_grads()
    fence()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    nbytes = flat.grads.numel() * 4
    ok &= report(rank, "flat_allreduce", exact and bool(torch.equal(lo, hi)), backend=dist.get_backend(), ranks=world, bytes=nbytes,
                 ms=ms, bus_GBps=2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9)
    del flat, p, idx, want

    # 2. the PathStorage reward exchange == one process replaying the global batch
    from interactron_amd.episode import _Adaptive
    from interactron_amd.storage import best_path_labels
    import numpy as np
    rng = np.random.default_rng(5)
    batches = []
    for b in (2 * world + 1, 2 * world, world + 1):
        batches.append((["root%d" % int(r) for r in rng.integers(0, 3, b)], torch.from_numpy(rng.integers(0, 4, (b, 5))),
                        [float(x) for x in rng.uniform(0.5, 3.0, b)]))

    class Host:
        path_storage = {}
        _dp_chunks = staticmethod(_Adaptive._dp_chunks)
        _dp_chunk_labels = _Adaptive._dp_chunk_labels

    host, single, got, want_labels, chunk = Host(), {}, [], [], 2
    for roots, actions, rewards in batches:
        data = {"frames": torch.zeros(len(roots), 5, 1), "actions": actions, "initial_image_path": roots}
        d = shard_batch(data, rank, world, by_root=True)
        b, mine = d["frames"].shape[0], d["dp_index"]
        for e0 in range(0, b, chunk):
            ep = list(range(e0, min(b, e0 + chunk)))
            got += list(zip([mine[t] for t in ep], host._dp_chunk_labels(d, e0 // chunk, chunk, ep, [rewards[mine[t]] for t in ep])))
        for c in range((b + chunk - 1) // chunk, host._dp_chunks(d, chunk)):
            host._dp_chunk_labels(d, c, chunk, [], [])
        labels = []
        for e0 in range(0, len(roots), chunk * world):
            sl = slice(e0, e0 + chunk * world)
            labels += best_path_labels(single, roots[sl], actions[sl, :4].tolist(), rewards[sl])
        want_labels += [(g, labels[g]) for g in mine]
    ok &= report(rank, "reward_exchange", sorted(got) == sorted(want_labels) and set(host.path_storage) == set(single), episodes=len(got))

    # 3. graph capture beside the collective's watchdog thread
    from interactron_amd import hipops as ops
    x = torch.randn(512, 256, device=dev)
    w = torch.randn(256, 256, device=dev)
    y_eager = ops.linear(ops.linear(x, w, None), w, None) if hasattr(ops, "linear") else (x @ w) @ w
    sx = x.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        sy = ops.linear(ops.linear(sx, w, None), w, None) if hasattr(ops, "linear") else (sx @ w) @ w
    g.replay()
    t = torch.ones(1024, device=dev)
    dist.all_reduce(t)
    g.replay()
    torch.cuda.synchronize()
    ok &= report(rank, "graph_capture_beside_watchdog", bool(torch.equal(sy, y_eager)) and float(t[0]) == world)

    # 4. two sharded meta-train steps of the real model, graphs on: replicas stay bit-identical
    import random
    from interactron_amd import Config, build_model, manual_seed
    from interactron_amd.synthetic import load_procedural, synthetic_episodes
    from interactron_amd.trainer import FlatOuterStep
    cfg = dict(TYPE="interactron", WEIGHTS="procedural", NUM_CLASSES=1235, SET_COST_CLASS=1.0, SET_COST_BBOX=5.0, SET_COST_GIOU=2.0,
               NUM_LAYERS=4, NUM_HEADS=8, EMBEDDING_DIM=512, BLOCK_SIZE=5 * (8 * 10 + 50) + 5, IMG_FEATURE_SIZE=256, OUTPUT_SIZE=512,
               BOX_EMB_SIZE=256, EMBEDDING_PDROP=0.1, RESIDUAL_PDROP=0.1, ATTENTION_PDROP=0.1, ADAPTIVE_LR=1e-3, EPISODE_CHUNK=2,
               STEP_GRAPH="on")
    model = build_model(Config(**cfg))
    load_procedural(model.fusion, "fusion.")
    model = model.to(dev).train()
    outer = FlatOuterStep(model, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
    manual_seed(42 + rank)
    random.seed(42 + rank)
    for step in range(3):
        data = synthetic_episodes(2 * world, height=128, width=160, tag="selftest%d" % step)
        d = shard_batch(data, rank, world, by_root=True)
        d["frames"], d["masks"] = d["frames"].to(dev), d["masks"].to(dev)
        d["category_ids"] = [[t.to(dev) for t in ep] for ep in d["category_ids"]]
        d["boxes"] = [[t.to(dev) for t in ep] for ep in d["boxes"]]
        model(d)
        outer.step()
    pchk = torch.stack([outer.flat.params.double().sum(), (outer.flat.params.double() ** 2).sum()])
    lo, hi = pchk.clone(), pchk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    kinds = sorted(type(v).__name__ for v in model.__dict__.get("_chunk_graphs", {}).values())
    ok &= report(rank, "sharded_steps_keep_replicas_identical", bool(torch.equal(lo, hi)) and bool(torch.isfinite(pchk).all()),
                 step_graphs=kinds)
    fence()
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--skip-bench", action="store_true")
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    sys.exit(worker(args) if args.worker else launcher(args))


if __name__ == "__main__":
    main()
