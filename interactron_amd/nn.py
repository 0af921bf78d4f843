"""Parameter-holding building blocks with the reference's attribute names, computing on the HIP kernels.

``torch.nn.Module`` is used purely as the parameter / state_dict container (so reference checkpoints load key for
key and ``utils.meta_utils.get_parameters``-style traversal sees the same tree); every ``forward`` goes through
``interactron_amd.hipops``.
"""
import math

import torch
from torch import nn

from . import hipops as ops


def _uniform(shape, bound):
    return nn.Parameter(torch.empty(shape).uniform_(-bound, bound))


class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        b = 1.0 / math.sqrt(in_features)
        self.in_features, self.out_features = in_features, out_features
        self.weight = _uniform((out_features, in_features), b)
        self.bias = _uniform((out_features,), b) if bias else None

    def forward(self, x, out_dtype=None):
        return ops.linear(x, self.weight, self.bias, out_dtype)


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


class Embedding(nn.Module):
    def __init__(self, num, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(num, dim))


class Dropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        return ops.dropout(x, self.p, self.training)


class PointwiseConv2d(nn.Module):
    """1x1 convolution with bias on NHWC tokens (DETR ``input_proj``, reference detr.py:40); weight kept as
    [out, in, 1, 1] for checkpoint compatibility."""

    def __init__(self, cin, cout):
        super().__init__()
        b = 1.0 / math.sqrt(cin)
        self.weight = _uniform((cout, cin, 1, 1), b)
        self.bias = _uniform((cout,), b)

    def forward(self, x):
        w = self.weight   # [out, in, 1, 1], or [E, out, in, 1, 1] for episode-batched fast weights
        return ops.linear(x, ops.weight_view(w, w.shape[:-2]), self.bias)


class MultiheadAttention(nn.Module):
    """nn.MultiheadAttention-compatible parameters (packed in_proj + out_proj child) on batch-first tokens.

    query [n, L, E], key/value [n, S, E]; ``key_padding_mask`` uint8/bool [n, S] (nonzero = ignore).
    Attention weights are materialised per (frame, head) as [n, H, L, S] (S padded to a multiple of 4) and go
    through the wave-per-row softmax; QK^T and PV are strided batched GEMMs straight out of the projection
    buffers (no head permutes)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, float(dropout)
        self.head_dim = embed_dim // num_heads
        b = math.sqrt(6.0 / (4 * embed_dim))
        self.in_proj_weight = _uniform((3 * embed_dim, embed_dim), b)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim)

    def forward(self, query, key, value, key_padding_mask=None, qk_same=False):
        n, L, E = query.shape
        S = key.shape[1]
        H, hd = self.num_heads, self.head_dim
        W, B = self.in_proj_weight, self.in_proj_bias
        if qk_same:   # self-attention with q = k input: one GEMM for both projections
            (Wqk, Wv), (Bqk, Bv) = ops.SplitRows.apply(W, 2 * E, E), ops.SplitRows.apply(B, 2 * E, E)
            qk = ops.linear(query, Wqk, Bqk)
            q, k, q_ld, k_ld, q_off, k_off = qk, qk, 2 * E, 2 * E, 0, E
        else:
            (Wq, Wk, Wv), (Bq, Bk, Bv) = ops.SplitRows.apply(W, E, E, E), ops.SplitRows.apply(B, E, E, E)
            q = ops.linear(query, Wq, Bq)
            k = ops.linear(key, Wk, Bk)
            q_ld, k_ld, q_off, k_off = E, E, 0, 0
        v = ops.linear(value, Wv, Bv)
        mask = None
        if key_padding_mask is not None:
            mask = key_padding_mask.to(torch.uint8).contiguous()
        o = ops.attention(q, k, v, n, H, L, S, hd, q_ld, k_ld, q_off, k_off, E, 0, 1.0 / math.sqrt(hd), mask,
                          self.dropout, self.training)
        return self.out_proj(o)


class FrozenBatchNorm2d(nn.Module):
    """Buffers only (reference backbone.py:19-54); folded once into a per-channel affine."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._fold = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        self._fold = None
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def folded(self):
        key = tuple((b.data_ptr(), b._version) for b in (self.weight, self.bias, self.running_mean, self.running_var))
        if self._fold is None or self._fold[0] != key:
            self._fold = (key, ops.bn_fold(self.weight, self.bias, self.running_mean, self.running_var, 1e-5))
        return self._fold[1]

    def forward(self, x, residual=None, relu=False):
        scale, shift = self.folded()
        return ops.BnAct.apply(x, scale, shift, residual, relu)


class Conv2dNHWC(nn.Module):
    """Bias-free convolution on NHWC activations.

    The parameter lives in HBM as [out, kh, kw, in] -- the column order of the patch matrix, so the contraction reads it
    (and every per-episode fast weight derived from it) as a plain k-contiguous operand with no per-call permute copy.
    ``state_dict`` / ``load_state_dict`` speak the reference's [out, in, kh, kw] layout (checkpoint compatibility)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1):
        super().__init__()
        self.stride, self.padding, self.dilation = stride, padding, dilation
        w = _uniform((cout, cin, k, k), math.sqrt(6.0 / (cin * k * k)))   # drawn in the reference's element order
        self.weight = nn.Parameter(w.data.permute(0, 2, 3, 1).contiguous())

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        w = destination[prefix + "weight"]
        destination[prefix + "weight"] = w.permute(0, 3, 1, 2)   # [out, in, kh, kw] view of the same storage

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "weight"
        w = state_dict.get(key)
        if w is not None and w.dim() == 4:
            state_dict[key] = w.permute(0, 2, 3, 1)
        try:
            super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        finally:
            if w is not None:
                state_dict[key] = w

    def forward(self, x):
        return ops.conv2d_nhwc(x, self.weight, self.stride, self.padding, self.dilation)
