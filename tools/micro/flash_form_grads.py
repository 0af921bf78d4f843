"""Per-tensor distance between the meta-gradients of one training step under the two tr forms of the flash kernels."""
import random
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from interactron_amd import hipops as ops  # noqa: E402
from test_parity_gpu import make, synthetic_episodes, to_gpu  # noqa: E402

H, W, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
res = {}
for form in ("bf16", "f16"):
    ops.FLASH_TR = form
    m = make("interactron")
    m.config.STEP_GRAPH = False
    data = to_gpu(synthetic_episodes(E, height=H, width=W, tag="dp"))
    random.seed(11)
    m.zero_grad()
    m(data)
    res[form] = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}
rel = {k: float((res["f16"][k] - res["bf16"][k]).norm() / res["bf16"][k].norm().clamp_min(1e-300)) for k in res["bf16"]}
for k, v in sorted(rel.items(), key=lambda kv: -kv[1])[:12]:
    print("%-60s %.2e   |g| %.2e" % (k, v, float(res["bf16"][k].norm())))
