"""Distance of predict()'s outputs (fixture G11: one adaptive step on 5 frames, then the frame-0 prediction) from the reference's
recorded ones under the four combinations of the contraction form (fp16x3 / bf16x6) and the flash kernels' tr form (f16 / bf16).
usage: python tools/form_noise.py"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from interactron_amd import _lib, hipops  # noqa: E402
from test_parity_gpu import make, synthetic_episodes, to_gpu  # noqa: E402


def main():
    import os
    M = torch.load(os.path.join("tests", "golden", "golden_model.pt"), weights_only=False)
    lib = _lib.load()
    m = make("interactron")
    ep = to_gpu(synthetic_episodes(1, tag="golden"))
    for x3 in (0, 1):
        for tr in ("bf16", "f16"):
            lib.ix_gemm_set_x3(x3)
            hipops.FLASH_TR = tr
            m.__dict__.setdefault("_predict_graphs", {})["disabled"] = True   # eager every time: a graph would replay the form it was captured with
            pred = m.predict(ep)
            out = []
            for k, rec in M["g11"].items():
                if "full" not in rec:
                    continue
                ref = rec["full"]
                d = (pred[k].detach().cpu().float() - ref).abs()
                out.append("%s max %.2e (%.2e of scale) rms %.2e" % (k, float(d.max()), float(d.max() / ref.abs().max()),
                                                                     float(d.pow(2).mean().sqrt())))
            print("contraction %s, flash tr %s: %s" % ("x3" if x3 else "x6", tr, "; ".join(out)))


if __name__ == "__main__":
    main()
