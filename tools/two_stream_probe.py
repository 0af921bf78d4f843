"""Would two half-batches on two streams beat one batch?  Two independent models (same weights), each stepping 8 episodes of
300 x 300 from its own thread on its own stream (captured graphs and all), against one model stepping 16.  Throughput only: the
two replicas do not share gradients.  usage (GPU box): python tools/two_stream_probe.py [steps]"""
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def make(E, tag, dev):
    from interactron_amd import Config, build_model
    from interactron_amd.synthetic import load_procedural, synthetic_episodes
    from interactron_amd.trainer import FlatOuterStep
    cfg, _ = bench.model_cfg(300, 50, E, "interactron")
    model = build_model(Config(**cfg))
    load_procedural(model.fusion, "fusion.")
    model = model.to(dev).train()
    outer = FlatOuterStep(model, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
    data = bench.to_gpu(synthetic_episodes(E, height=300, width=300, tag=tag), dev)

    def step():
        model(data)
        outer.step()
    return step


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    random.seed(1)
    one = make(16, "one", dev)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    t_one = (time.perf_counter() - t0) / steps
    print("one model, 16 episodes per step: %.1f ms per step, %.1f frames/s" % (t_one * 1e3, 80 / t_one))
    del one
    torch.cuda.empty_cache()

    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    fns = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            f = make(8, "half%d" % i, dev)
            for _ in range(3):
                f()
        st.synchronize()
        fns.append(f)

    def worker(f, st, n):
        with torch.cuda.stream(st):
            for _ in range(n):
                f()
        st.synchronize()

    for label, conc in (("one after the other", False), ("two threads, two streams", True)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if conc:
            th = [threading.Thread(target=worker, args=(f, st, steps)) for f, st in zip(fns, streams)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        else:
            for f, st in zip(fns, streams):
                worker(f, st, steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("two models, 8 episodes each, %s: %.1f ms per pair of steps, %.1f frames/s" % (label, dt * 1e3, 80 / dt))


if __name__ == "__main__":
    main()
