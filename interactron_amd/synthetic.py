"""RNG-free procedural weights and synthetic episodes.

Real Interactron weights / data are S3 tarballs that are not available offline
(reference README.md:23-24).  Every parity fixture, test and benchmark in this
repo therefore uses a closed-form, counter-hash initialiser keyed by the
parameter *name* (so that the imported reference, the CPU oracle and the HIP
build regenerate bit-identical tensors without shipping a 235 MB state_dict)
and synthetic 5-frame episodes laid out exactly like the reference's
``collate_fn`` output (reference utils/storage_utils.py:53-64).

Nothing here depends on torch's RNG stream: values come from a splitmix64
counter hash evaluated in numpy uint64 arithmetic, which is exact on every
platform.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def hash_uniform(tag, n, lo=0.0, hi=1.0):
    """n float64 samples in [lo, hi) that depend only on (tag, index)."""
    seed = np.uint64(zlib.crc32(tag.encode("utf-8")))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (seed << np.uint64(32))
        bits = _splitmix64(_splitmix64(idx))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def hash_normal(tag, n):
    """Box-Muller on two hashed uniform streams (float64)."""
    u1 = hash_uniform(tag + "#a", n)
    u2 = hash_uniform(tag + "#b", n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def hash_randint(tag, n, lo, hi):
    """n int64 samples uniform on {lo..hi} inclusive."""
    return (lo + np.floor(hash_uniform(tag, n) * (hi - lo + 1))).astype(np.int64)


def _centered(tag, shape, half_width, mean=0.0):
    n = int(np.prod(shape)) if len(shape) else 1
    v = mean + (2.0 * hash_uniform(tag, n) - 1.0) * half_width
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def procedural_tensor(name, shape):
    """Closed-form value for the state_dict entry ``name`` with ``shape``.

    Scales are fan-in based so a 50-conv / 16-transformer-layer stack keeps O(1)
    activations without trained BatchNorm statistics:
      * conv / linear weights: uniform, std = gain / sqrt(fan_in)
        (gain sqrt(2) in the ResNet body, 1 elsewhere)
      * the last FrozenBN of every bottleneck (bn3) has weight ~0.3 so residual
        branches do not blow the trunk up
      * biases, LayerNorm/BN shifts: small; LayerNorm/BN scales: 1 +- 0.1
    """
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "pos_embed" and shape == (1, 1805, 512):
        # fixed sin/cos table of the decoder-style fusion (reference new_transformer.py:62-73): keep the module's own
        return None
    is_bn = ".bn" in name or "downsample.1" in name
    if is_bn:
        if leaf == "weight":
            return _centered(name, shape, 0.03, 0.3) if ".bn3." in name else _centered(name, shape, 0.1, 1.0)
        if leaf == "running_var":
            return _centered(name, shape, 0.1, 1.0)
        return _centered(name, shape, 0.05)  # bias, running_mean
    if "norm" in name or ".ln" in name or "ln_f" in name:
        return _centered(name, shape, 0.1, 1.0) if leaf == "weight" else _centered(name, shape, 0.05)
    if leaf == "empty_weight":
        w = torch.ones(shape)
        w[-1] = 0.1
        return w
    if leaf == "mask" and len(shape) == 4:  # GPT attention mask buffer: all ones (reference models/gpt.py:35-36)
        return torch.ones(shape)
    if name.endswith("query_embed.weight"):
        return _centered(name, shape, 1.0)
    if leaf in ("seq_pos_embed", "pos_emb", "action_tokens", "query_embed"):
        return _centered(name, shape, 0.05)
    if leaf in ("bias", "in_proj_bias"):
        return _centered(name, shape, 0.02)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        gain = np.sqrt(2.0) if "backbone" in name else 1.0
        return _centered(name, shape, float(gain * np.sqrt(3.0 / fan_in)))
    return _centered(name, shape, 0.02)


def procedural_state_dict(shapes):
    """``shapes``: mapping name -> shape (e.g. from ``module.state_dict()``)."""
    out = {}
    for k, v in shapes.items():
        t = procedural_tensor(k, tuple(v.shape) if hasattr(v, "shape") else tuple(v))
        if t is not None:
            out[k] = t
    return out


def load_procedural(module, prefix=""):
    """Overwrite every parameter and buffer of ``module`` in place (keeps device)."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        t = procedural_tensor(prefix + k, tuple(v.shape))
        new[k] = v if t is None else t.to(dtype=v.dtype)
    module.load_state_dict(new, strict=True)
    return module


def synthetic_episodes(batch, frames=5, height=300, width=300, tag="bench", device="cpu"):
    """Synthetic ``data`` dict in the reference's collate layout.

    frames ~ N(0,1) (stands in for ImageNet-normalised pixels), masks int64
    zeros, frame s has 3+s ground-truth boxes with labels U{1..1234},
    cx,cy ~ U(0.3,0.7), w,h ~ U(0.05,0.35), actions ~ U{0..3}.
    """
    n = batch * frames * 3 * height * width
    fr = torch.from_numpy(hash_normal(tag + "/frames", n).astype(np.float32)).reshape(batch, frames, 3, height, width)
    data = {
        "frames": fr.to(device),
        "masks": torch.zeros(batch, frames, height, width, dtype=torch.long, device=device),
        "actions": torch.from_numpy(hash_randint(tag + "/actions", batch * frames, 0, 3)).reshape(batch, frames).to(device),
        "category_ids": [],
        "boxes": [],
        "episode_ids": torch.arange(batch * frames, dtype=torch.long).reshape(batch, frames),
        "initial_image_path": ["%s/ep%d" % (tag, i) for i in range(batch)],
    }
    for b in range(batch):
        cats, boxes = [], []
        for s in range(frames):
            k = 3 + s
            t = "%s/ep%d/f%d" % (tag, b, s)
            cats.append(torch.from_numpy(hash_randint(t + "/cls", k, 1, 1234)).to(device))
            cxcy = hash_uniform(t + "/c", 2 * k, 0.3, 0.7).reshape(k, 2)
            wh = hash_uniform(t + "/wh", 2 * k, 0.05, 0.35).reshape(k, 2)
            boxes.append(torch.from_numpy(np.concatenate([cxcy, wh], 1).astype(np.float32)).to(device))
        data["category_ids"].append(cats)
        data["boxes"].append(boxes)
    return data


def evalrun_weight_edit(state_dict, overrides=None, sharpen=8.0, query_gain=4.0, loss_gain=5e-4):
    """Fixture G18's weight recipe (tests/golden/make_golden_evalrun.py), applied in place to an ``interactron`` state
    dict on BOTH sides (the imported reference there, this package in tests/test_engine_gpu.py).

    With procedural weights the detector's 50 queries are indistinguishable (their outputs differ by 3e-4 of their
    magnitude): every query predicts the same arbitrary class and box, one detection per image survives NMS and every AP is
    zero.  Closed-form part: the decoder's cross-attention query / key projections x `sharpen` and the learned queries
    x `query_gain` -- attention that actually selects tokens, as in a trained DETR; queries then differ by 5 % -- and the
    learned-loss head's last layer x `loss_gain`: with procedural weights the learned loss' gradient clips the inner step at
    +-0.01 on nearly every detector weight, an adaptation that replaces the detector instead of adjusting it.
    `overrides` (stored IN the fixture, computed there from a calibration pass of the reference): rows of the class head for
    the dataset's categories and the last layer of the box head re-centred on the mean query so that classes, scores and boxes
    spread over the queries, and the policy head's bias minus its mean output so that the chosen move depends on the frames.
    Everything downstream -- adaptation step, boxes, scores, NMS survivors, matching, policy -- is what the networks
    compute.  An override is a full tensor or ``{"rows": LongTensor, "values": Tensor}``."""
    for k, v in state_dict.items():
        if ".decoder.layers." in k and k.endswith("multihead_attn.in_proj_weight"):
            v[:2 * v.shape[1]] *= sharpen
    state_dict["detector.query_embed.weight"] *= query_gain
    state_dict["fusion.loss_decoder.layers.2.weight"] *= loss_gain
    state_dict["fusion.loss_decoder.layers.2.bias"] *= loss_gain
    for k, o in (overrides or {}).items():
        if isinstance(o, dict):
            state_dict[k][o["rows"].to(state_dict[k].device)] = o["values"].to(state_dict[k])
        else:
            state_dict[k].copy_(o)
    return state_dict
