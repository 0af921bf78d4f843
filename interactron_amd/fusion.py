"""Fusion transformers over the multi-frame rollout, on the HIP kernels.

``Transformer``        -- GPT-style fusion of configs ``multi_frame_baseline`` / ``interactron``
                          (reference models/transformer.py:33-66 + models/gpt.py:13-200)
``DecoderTransformer`` -- DETR-decoder-style fusion of config ``interactron_random``
                          (reference models/new_transformer.py:12-73)
Attribute names / child order reproduce the reference's ``state_dict`` keys (``fusion.model.blocks.0.attn.key.weight``,
``fusion.model.seq_pos_embed``, ``fusion.action_tokens`` ...).
"""
import math

import numpy as np
import torch
from torch import nn

from . import hipops as ops
from .detector import MLP, TransformerDecoder
from .nn import Dropout, LayerNorm, Linear


class SelfAttention(nn.Module):
    """reference gpt.py:13-57: separate key/query/value projections, full (non-causal) attention -- the reference's
    mask buffer is all ones -- dropout on the attention weights and on the projected output."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg.EMBEDDING_DIM
        assert d % cfg.NUM_HEADS == 0
        self.key = Linear(d, d)
        self.query = Linear(d, d)
        self.value = Linear(d, d)
        self.attn_drop = Dropout(cfg.ATTENTION_PDROP)
        self.resid_drop = Dropout(cfg.RESIDUAL_PDROP)
        self.proj = Linear(d, d)
        if cfg.BLOCK_SIZE <= 4096:   # kept only for checkpoint-key compatibility; never read (all ones == no masking)
            self.register_buffer("mask", torch.ones(1, 1, cfg.BLOCK_SIZE, cfg.BLOCK_SIZE))
        self.NUM_HEADS = cfg.NUM_HEADS

    def forward(self, x):
        B, T, C = x.shape
        H = self.NUM_HEADS
        hd = C // H
        # the three projections of the same input as ONE contraction (N = 3C instead of three N = C launches, one
        # input-gradient contraction and no gradient sums over x); the attention kernels read [k | q | v] in place and hand
        # the gradient back as one buffer of that layout
        w = ops.CatParams.apply(self.key.weight, self.query.weight, self.value.weight)
        b = ops.CatParams.apply(self.key.bias, self.query.bias, self.value.bias)
        kqv = ops.linear(x, w, b)
        y = ops.attention(kqv, kqv, kqv, B, H, T, T, hd, 3 * C, 3 * C, C, 0, 3 * C, 2 * C, 1.0 / math.sqrt(hd), None,
                          self.attn_drop.p, self.attn_drop.training)
        return self.proj(y)   # (resid_drop is applied fused with the residual add in Block.forward)


class _Gelu(nn.Module):
    def forward(self, x):
        return ops.Gelu.apply(x)


class Block(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.EMBEDDING_DIM
        self.ln1 = LayerNorm(d)
        self.ln2 = LayerNorm(d)
        self.attn = SelfAttention(cfg)
        self.mlp = nn.Sequential(Linear(d, 4 * d), _Gelu(), Linear(4 * d, d), Dropout(cfg.RESIDUAL_PDROP))

    def forward(self, x):
        # x + dropout(branch) as one pass each (reference gpt.py:75-77; the dropouts are attn.resid_drop and mlp[3])
        x, r = ops.fanout(x, 2)   # (branch + residual: one gradient sum, hipops.Fanout)
        x, r = ops.fanout(ops.add_dropout(r, self.attn(self.ln1(x)), self.attn.resid_drop.p, self.attn.resid_drop.training), 2)
        m = self.mlp[2](self.mlp[1](self.mlp[0](self.ln2(x))))
        return ops.add_dropout(r, m, self.mlp[3].p, self.mlp[3].training)


class GPT(nn.Module):
    """reference gpt.py:81-200 (learned position table sized BLOCK_SIZE; 2060 in the shipped configs)."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg.EMBEDDING_DIM
        self.pos_emb = nn.Parameter(torch.zeros(1, 255, d))                 # present in checkpoints, never used
        self.seq_pos_embed = nn.Parameter(torch.zeros(1, cfg.BLOCK_SIZE, d))
        self.embed_dim = d
        self.drop = Dropout(cfg.EMBEDDING_PDROP)
        self.blocks = nn.Sequential(*[Block(cfg) for _ in range(cfg.NUM_LAYERS)])
        self.ln_f = LayerNorm(d)
        self.head = Linear(d, cfg.OUTPUT_SIZE, bias=False)
        self.block_size = cfg.BLOCK_SIZE
        for m in self.modules():
            if isinstance(m, Linear):
                m.weight.data.normal_(mean=0.0, std=0.02)
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, seq):
        b, t, d = seq.shape
        assert t <= self.block_size, "Cannot forward, model block size is exhausted."
        pe = self.seq_pos_embed[:, :t, :].reshape(t * d)
        x = self.drop(ops.AddRowVec.apply(seq.reshape(b, t * d), pe, 1).reshape(b, t, d))
        x = self.blocks(x)
        return self.head(self.ln_f(x))


def _like(t, ref):
    """t in ref's storage dtype (16-bit mode: the detector's fp32 head outputs and the fp32 token parameters join bf16 sequences)"""
    if t.dtype == ref.dtype:
        return t
    from . import b16
    return b16.to_b16(t) if ref.dtype == torch.bfloat16 else b16.to_f32(t)


def _token_inputs(mod, x):
    emf, bf = x["embedded_memory_features"], x["box_features"]
    if ops.b16_fusion_only() and emf.dtype == torch.float32:
        # MODEL.COMPUTE_DTYPE bf16_fusion: the detector ran fp32-grade; from here to the decoders activations are bf16 (the conversions are
        # differentiable: the gradient re-enters the detector as fp32)
        from . import b16
        emf, bf = b16.to_b16(emf), b16.to_b16(bf)
    img = mod.img_feature_embedding(emf.permute(0, 1, 3, 4, 2))
    if bf.dtype == torch.bfloat16:
        # 16-bit mode: the prediction embedding reads the detector's fp32 logits (|logit| up to ~10: bf16 would round them by 0.02-0.04)
        # and is tiny (50 rows per frame) -- it runs fp32-grade, its output joins the bf16 token sequence
        from . import b16
        preds = torch.cat((b16.to_f32(bf), x["pred_logits"], x["pred_boxes"]), dim=-1)
        return img, b16.to_b16(mod.prediction_embedding(preds))
    preds = torch.cat((bf, x["pred_logits"], x["pred_boxes"]), dim=-1)
    return img, mod.prediction_embedding(preds)


def _decode(mod, y_preds, y_actions):
    y_seq, y_box, y_logit, y_loss = ops.fanout(y_preds, 4)
    return {"seq": y_seq.squeeze(),
            "pred_boxes": ops.Sigmoid.apply(mod.box_decoder(y_box)).squeeze(),
            "pred_logits": mod.logit_decoder(y_logit, out_dtype=torch.float32).squeeze(),
            "loss": mod.loss_decoder(y_loss),
            "actions": mod.action_decoder(y_actions).squeeze()}


class Transformer(nn.Module):
    """reference models/transformer.py:33-66."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg.EMBEDDING_DIM
        self.img_feature_embedding = Linear(cfg.IMG_FEATURE_SIZE, d)
        self.prediction_embedding = Linear(cfg.BOX_EMB_SIZE + cfg.NUM_CLASSES + 5, d)
        self.model = GPT(cfg)
        self.box_decoder = MLP(cfg.OUTPUT_SIZE, 256, 4, 3)
        self.logit_decoder = Linear(cfg.OUTPUT_SIZE, cfg.NUM_CLASSES + 1)
        self.loss_decoder = MLP(cfg.OUTPUT_SIZE, 512, 1, 3)
        self.action_decoder = MLP(cfg.OUTPUT_SIZE, 512, 4, 3)
        self.action_tokens = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(1, 5, d), a=math.sqrt(5)))

    def forward(self, x):
        img, pred = _token_inputs(self, x)
        b, s, p, n = pred.shape
        n_preds = s * p
        seq = torch.cat((img.reshape(b, -1, n), pred.reshape(b, -1, n), _like(self.action_tokens, pred).repeat(b, 1, 1)), dim=1)
        y = self.model(seq)
        y_preds = y[:, -(n_preds + 5):-5].reshape(b, s, p, -1)
        return _decode(self, y_preds, y[:, -5:-1].reshape(b, 4, -1))


def _sincos_1d(dim, pos):
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    out = np.einsum("m,d->md", np.asarray(pos, dtype=np.float64).reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def fixed_memory_pos_embed(embed_dim, grid, frames=5):
    """Fixed 2-D sin/cos (image position) + 1-D sin/cos (frame index) table, reference new_transformer.py:62-73."""
    gh, gw = np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)
    half = embed_dim // 2
    img = np.concatenate([_sincos_1d(half // 2, g[0]), _sincos_1d(half // 2, g[1])], axis=1)
    img_pos = torch.zeros(1, grid * grid, embed_dim)
    img_pos[:, :, :half] = torch.from_numpy(img).float()
    seq_pos = torch.zeros(1, frames, embed_dim)
    seq_pos[:, :, half:] = torch.from_numpy(_sincos_1d(half, np.arange(frames))).float()
    pos = torch.zeros(1, frames * grid * grid, embed_dim)
    for i in range(frames):
        pos[:, grid * grid * i:grid * grid * (i + 1)] = img_pos + seq_pos[:, i]
    return pos


class DecoderTransformer(nn.Module):
    """reference models/new_transformer.py:12-60: 255 target tokens (250 predictions + 5 action tokens) attend to
    1805 memory tokens through 4 DETR decoder layers (d=512)."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg.EMBEDDING_DIM
        self.img_feature_embedding = Linear(cfg.IMG_FEATURE_SIZE, d)
        self.prediction_embedding = Linear(cfg.BOX_EMB_SIZE + cfg.NUM_CLASSES + 5, d)
        self.box_decoder = MLP(cfg.OUTPUT_SIZE, 512, 4, 3)
        self.logit_decoder = Linear(cfg.OUTPUT_SIZE, cfg.NUM_CLASSES + 1)
        self.loss_decoder = MLP(cfg.OUTPUT_SIZE, 512, 1, 3)
        self.action_decoder = MLP(cfg.OUTPUT_SIZE, 512, 4, 3)
        self.action_tokens = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(1, 5, d), a=math.sqrt(5)))
        self.transformer = TransformerDecoder(d, cfg.NUM_HEADS, 2048, 0.1, cfg.NUM_LAYERS)
        self.embed_dim = d
        self.img_len = 19 * 19
        self.pos_embed = nn.Parameter(fixed_memory_pos_embed(d, 19), requires_grad=False)
        self.query_embed = nn.Parameter(torch.zeros(1, 255, d), requires_grad=True)

    def forward(self, x):
        img, pred = _token_inputs(self, x)
        b, s, p, n = pred.shape
        # (the reference indexes the decoder output as [1, tokens, ...], i.e. b == 1; b > 1 = independent sequences, used
        #  by the episode-batched training step)
        dev = pred.device
        mem_len, L = 5 * self.img_len, self.img_len
        parts = [img.reshape(b, -1, n)]
        if s * L < mem_len:
            parts.append(torch.zeros(b, mem_len - s * L, n, device=dev, dtype=img.dtype))
        memory = torch.cat(parts, dim=1)
        parts = [pred.reshape(b, -1, n)]
        if s * p < 250:
            parts.append(torch.zeros(b, 250 - s * p, n, device=dev, dtype=pred.dtype))
        parts.append(_like(self.action_tokens, pred).repeat(b, 1, 1))
        tgt = torch.cat(parts, dim=1)
        y = self.transformer(tgt, memory, None, self.pos_embed, self.query_embed.reshape(255 * n))
        y_preds = y[:, :-5].reshape(b, s, p, -1)
        return _decode(self, y_preds, y[:, -5:-1].reshape(b, 4, -1))
