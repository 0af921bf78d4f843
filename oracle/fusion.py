"""Oracle: the two fusion transformers (GPT style and decoder style).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Keys of ``sd`` are the
reference's ``fusion.*`` checkpoint names with the ``fusion.`` prefix removed.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .detector import decoder_layer, layer_norm, mlp


def _lin(x, sd, p):
    return F.linear(x, sd[p + "weight"], sd.get(p + "bias"))


def gpt_attention(x, sd, p, nhead, pdrop, training):
    """Full (non-causal: the mask buffer is all ones) self-attention, reference gpt.py:39-57."""
    B, T, C = x.shape
    hd = C // nhead
    k = _lin(x, sd, p + "key.").view(B, T, nhead, hd).transpose(1, 2)
    q = _lin(x, sd, p + "query.").view(B, T, nhead, hd).transpose(1, 2)
    v = _lin(x, sd, p + "value.").view(B, T, nhead, hd).transpose(1, 2)
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
    att = F.dropout(F.softmax(att, dim=-1), pdrop, training)
    y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
    return F.dropout(_lin(y, sd, p + "proj."), pdrop, training)


def gpt_forward(seq, sd, cfg, p="model.", training=False):
    """GPT.forward (reference gpt.py:189-200): learned position table, pre-norm blocks, ln_f, bias-free head."""
    t = seq.shape[1]
    assert t <= cfg["BLOCK_SIZE"]
    x = F.dropout(seq + sd[p + "seq_pos_embed"][:, :t, :], cfg["EMBEDDING_PDROP"], training)
    for i in range(cfg["NUM_LAYERS"]):
        b = "%sblocks.%d." % (p, i)
        x = x + gpt_attention(layer_norm(x, sd, b + "ln1."), sd, b + "attn.", cfg["NUM_HEADS"],
                              cfg["ATTENTION_PDROP"], training)
        h = F.gelu(_lin(layer_norm(x, sd, b + "ln2."), sd, b + "mlp.0."))
        x = x + F.dropout(_lin(h, sd, b + "mlp.2."), cfg["RESIDUAL_PDROP"], training)
    return F.linear(layer_norm(x, sd, p + "ln_f."), sd[p + "head.weight"])


def _embed_inputs(x, sd):
    img = _lin(x["embedded_memory_features"].permute(0, 1, 3, 4, 2), sd, "img_feature_embedding.")
    preds = torch.cat((x["box_features"], x["pred_logits"], x["pred_boxes"]), dim=-1)
    return img, _lin(preds, sd, "prediction_embedding.")


def _decode(y_preds, y_act, sd):
    return {
        "seq": y_preds.squeeze(),
        "pred_boxes": mlp(y_preds, sd, "box_decoder.", 3).sigmoid().squeeze(),
        "pred_logits": _lin(y_preds, sd, "logit_decoder.").squeeze(),
        "loss": mlp(y_preds, sd, "loss_decoder.", 3),
        "actions": mlp(y_act, sd, "action_decoder.", 3).squeeze(),
    }


def fusion_gpt_forward(sd, x, cfg, training=False):
    """models/transformer.py:47-66.  x: detector outputs with a leading batch dim of 1."""
    img, pred = _embed_inputs(x, sd)
    b, s, p, n = pred.shape
    seq = torch.cat((img.reshape(b, -1, n), pred.reshape(b, -1, n), sd["action_tokens"].repeat(b, 1, 1)), dim=1)
    y = gpt_forward(seq, sd, cfg, training=training)
    y_preds = y[:, -(s * p + 5):-5].reshape(b, s, p, -1)
    return _decode(y_preds, y[:, -5:-1].reshape(b, 4, -1), sd)


def sincos_1d(dim, pos):
    """reference new_transformer.py:109-129 (np.float -> float64)."""
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    out = np.einsum("m,d->md", np.asarray(pos, dtype=np.float64).reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def decoder_fusion_pos_embed(embed_dim=512, grid=19):
    """Fixed memory position table of the decoder-style fusion (reference new_transformer.py:62-73)."""
    gh, gw = np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)
    half = embed_dim // 2
    img = np.concatenate([sincos_1d(half // 2, g[0]), sincos_1d(half // 2, g[1])], axis=1)
    img_pos = torch.zeros(1, grid * grid, embed_dim)
    img_pos[:, :, :half] = torch.from_numpy(img).float()
    seq_pos = torch.zeros(1, 5, embed_dim)
    seq_pos[:, :, half:] = torch.from_numpy(sincos_1d(half, np.arange(5))).float()
    pos = torch.zeros(1, 5 * grid * grid, embed_dim)
    for i in range(5):
        pos[:, grid * grid * i:grid * grid * (i + 1)] = img_pos + seq_pos[:, i]
    return pos


def fusion_decoder_forward(sd, x, cfg, training=False):
    """models/new_transformer.py:33-60: 4 DETR decoder layers, tgt 255 tokens x memory 1805 tokens."""
    img, pred = _embed_inputs(x, sd)
    b, s, p, n = pred.shape
    memory = torch.zeros(b, 5 * 19 * 19, n)
    memory[:, :s * 19 * 19] = img.reshape(b, -1, n)
    tgt = torch.zeros(b, 255, n)
    tgt[:, :s * 50] = pred.reshape(b, -1, n)
    tgt[:, 250:255] = sd["action_tokens"].repeat(b, 1, 1)
    mask = torch.zeros(b, 5 * 19 * 19, dtype=torch.bool)
    t, mem = tgt.permute(1, 0, 2), memory.permute(1, 0, 2)
    pos, qpos = sd["pos_embed"].permute(1, 0, 2), sd["query_embed"].permute(1, 0, 2)
    for i in range(cfg["NUM_LAYERS"]):
        t = decoder_layer(t, mem, pos, qpos, mask, sd, "transformer.layers.%d." % i, cfg["NUM_HEADS"], 0.1, training)
    y = layer_norm(t, sd, "transformer.norm.").unsqueeze(0)  # TransformerDecoder returns output.unsqueeze(0): [1,255,b,n]
    # reference indexes this [1,255,b,n] tensor as if it were [b,255,n] (b == 1): y[:, :-5] keeps tokens 0..249
    y_preds = y[:, :-5].reshape(b, s, p, -1)
    return _decode(y_preds, y[:, -5:-1].reshape(b, 4, -1), sd)


def fusion_state_shapes(cfg, style="gpt"):
    """Ordered name -> shape of the fusion module's state_dict (reference models/transformer.py:35-45)."""
    d, o, c = cfg["EMBEDDING_DIM"], cfg["OUTPUT_SIZE"], cfg["NUM_CLASSES"]
    out = {}

    def lin(p, oo, ii, bias=True):
        out[p + "weight"] = (oo, ii)
        if bias:
            out[p + "bias"] = (oo,)

    def ln(p):
        out[p + "weight"] = (d,)
        out[p + "bias"] = (d,)

    def mlp3(p, i, h, oo):
        lin(p + "layers.0.", h, i)
        lin(p + "layers.1.", h, h)
        lin(p + "layers.2.", oo, h)

    out["action_tokens"] = (1, 5, d)
    if style == "decoder":
        out["pos_embed"] = (1, 1805, d)
        out["query_embed"] = (1, 255, d)
    lin("img_feature_embedding.", d, cfg["IMG_FEATURE_SIZE"])
    lin("prediction_embedding.", d, cfg["BOX_EMB_SIZE"] + c + 5)
    if style == "gpt":
        out["model.pos_emb"] = (1, 255, d)
        out["model.seq_pos_embed"] = (1, cfg["BLOCK_SIZE"], d)   # (2060 in the shipped configs, reference gpt.py:118-119)
        for i in range(cfg["NUM_LAYERS"]):
            b = "model.blocks.%d." % i
            ln(b + "ln1.")
            ln(b + "ln2.")
            if cfg["BLOCK_SIZE"] <= 4096:   # all-ones buffer, never read (gpt.py:35-36); at 800x800 it would be 650 MB a layer
                out[b + "attn.mask"] = (1, 1, cfg["BLOCK_SIZE"], cfg["BLOCK_SIZE"])
            for nm in ("key", "query", "value", "proj"):
                lin("%sattn.%s." % (b, nm), d, d)
            lin(b + "mlp.0.", 4 * d, d)
            lin(b + "mlp.2.", d, 4 * d)
        ln("model.ln_f.")
        lin("model.head.", o, d, bias=False)
        mlp3("box_decoder.", o, 256, 4)
    else:
        mlp3("box_decoder.", o, 512, 4)
    lin("logit_decoder.", c + 1, o)
    mlp3("loss_decoder.", o, 512, 1)
    mlp3("action_decoder.", o, 512, 4)
    if style == "decoder":
        for i in range(cfg["NUM_LAYERS"]):
            p = "transformer.layers.%d." % i
            for a in ("self_attn.", "multihead_attn."):
                out[p + a + "in_proj_weight"] = (3 * d, d)
                out[p + a + "in_proj_bias"] = (3 * d,)
                lin(p + a + "out_proj.", d, d)
            lin(p + "linear1.", 2048, d)
            lin(p + "linear2.", d, 2048)
            ln(p + "norm1.")
            ln(p + "norm2.")
            ln(p + "norm3.")
        ln("transformer.norm.")
    return out
