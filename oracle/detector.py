"""Oracle: DETR ResNet-50-DC5 detector as pure functions over a state_dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  ``sd`` maps the reference's
checkpoint key names (``backbone.0.body.layer2.0.conv1.weight`` ...) to CPU
tensors; differentiating w.r.t. a weight just means that entry requires grad.
"""
import math

import torch
import torch.nn.functional as F

RESNET50_BLOCKS = (3, 4, 6, 3)
# torchvision resnet50(replace_stride_with_dilation=[False, False, True]):
# layer4 keeps stride 1 and runs its 3x3 convs with dilation 2, except the first
# block which keeps the previous dilation 1 (reference backbone.py:88-90).
LAYER_CFG = (
    # name, planes, stride of first block, dilation of first block, dilation of the rest
    ("layer1", 64, 1, 1, 1),
    ("layer2", 128, 2, 1, 1),
    ("layer3", 256, 2, 1, 1),
    ("layer4", 512, 1, 1, 2),
)


def frozen_bn(x, sd, p):
    """FrozenBatchNorm2d, eps inside the rsqrt (reference backbone.py:44-54)."""
    scale = sd[p + "weight"] * (sd[p + "running_var"] + 1e-5).rsqrt()
    shift = sd[p + "bias"] - sd[p + "running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


def bottleneck(x, sd, p, stride, dilation):
    """ResNet v1.5 bottleneck: stride lives on the 3x3 conv."""
    y = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
    y = F.conv2d(y, sd[p + "conv2.weight"], stride=stride, padding=dilation, dilation=dilation)
    y = F.relu(frozen_bn(y, sd, p + "bn2."))
    y = frozen_bn(F.conv2d(y, sd[p + "conv3.weight"]), sd, p + "bn3.")
    if (p + "downsample.0.weight") in sd:
        x = frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1.")
    return F.relu(y + x)


def resnet50_dc5(x, sd, p="backbone.0.body."):
    """[n,3,H,W] -> layer4 features [n,2048,ceil(H/16),ceil(W/16)]."""
    x = F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)
    x = F.relu(frozen_bn(x, sd, p + "bn1."))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for (name, _planes, stride0, dil0, dil), nblocks in zip(LAYER_CFG, RESNET50_BLOCKS):
        for b in range(nblocks):
            x = bottleneck(x, sd, "%s%s.%d." % (p, name, b), stride0 if b == 0 else 1, dil0 if b == 0 else dil)
    return x


def downsample_mask(mask, size):
    """Nearest-neighbour mask resize (reference backbone.py:77)."""
    return F.interpolate(mask[None].float(), size=size).to(torch.bool)[0]


def sine_position(mask, num_pos_feats=128, temperature=10000.0):
    """PositionEmbeddingSine(normalize=True) (reference position_encoding.py:28-48)."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    scale = 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + 1e-6) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + 1e-6) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    px = x_embed[:, :, :, None] / dim_t
    py = y_embed[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def layer_norm(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def multihead_attention(query, key, value, sd, p, nhead, key_padding_mask=None, dropout_p=0.0, training=False):
    """nn.MultiheadAttention forward on [L, N, E] tensors, packed in_proj.

    q is scaled by 1/sqrt(head_dim) before QK^T, an additive -inf mask hides padded
    keys, dropout acts on the attention weights (torch F.multi_head_attention_forward).
    """
    L, N, E = query.shape
    S = key.shape[0]
    hd = E // nhead
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(query, w[:E], b[:E]).reshape(L, N * nhead, hd).transpose(0, 1)
    k = F.linear(key, w[E:2 * E], b[E:2 * E]).reshape(S, N * nhead, hd).transpose(0, 1)
    v = F.linear(value, w[2 * E:], b[2 * E:]).reshape(S, N * nhead, hd).transpose(0, 1)
    q = q * math.sqrt(1.0 / float(hd))
    att = torch.bmm(q, k.transpose(1, 2))
    if key_padding_mask is not None:
        add = torch.zeros(N, 1, 1, S).masked_fill(key_padding_mask.reshape(N, 1, 1, S), float("-inf"))
        att = (att.reshape(N, nhead, L, S) + add).reshape(N * nhead, L, S)
    att = F.dropout(att.softmax(-1), dropout_p, training)
    out = torch.bmm(att, v).transpose(0, 1).reshape(L, N, E)
    return F.linear(out, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _drop(x, p, training):
    return F.dropout(x, p, training)


def encoder_layer(src, pos, mask, sd, p, nhead=8, dropout=0.1, training=False):
    """Post-norm encoder layer (reference transformer.py:148-161)."""
    qk = src + pos
    a = multihead_attention(qk, qk, src, sd, p + "self_attn.", nhead, mask, dropout, training)
    src = layer_norm(src + _drop(a, dropout, training), sd, p + "norm1.")
    f = F.linear(_drop(F.relu(F.linear(src, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), dropout, training),
                 sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return layer_norm(src + _drop(f, dropout, training), sd, p + "norm2.")


def decoder_layer(tgt, memory, pos, query_pos, mask, sd, p, nhead=8, dropout=0.1, training=False):
    """Post-norm decoder layer (reference transformer.py:211-232)."""
    qk = tgt + query_pos
    a = multihead_attention(qk, qk, tgt, sd, p + "self_attn.", nhead, None, dropout, training)
    tgt = layer_norm(tgt + _drop(a, dropout, training), sd, p + "norm1.")
    c = multihead_attention(tgt + query_pos, memory + pos, memory, sd, p + "multihead_attn.", nhead, mask, dropout,
                            training)
    tgt = layer_norm(tgt + _drop(c, dropout, training), sd, p + "norm2.")
    f = F.linear(_drop(F.relu(F.linear(tgt, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), dropout, training),
                 sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return layer_norm(tgt + _drop(f, dropout, training), sd, p + "norm3.")


def detr_transformer(src, mask, query_embed, pos, sd, p="transformer.", layers=6, training=False):
    """reference transformer.py:46-58; only the last decoder layer's normed output is used (detr.py:69)."""
    n, c, h, w = src.shape
    x = src.flatten(2).permute(2, 0, 1)
    pe = pos.flatten(2).permute(2, 0, 1)
    qe = query_embed.unsqueeze(1).repeat(1, n, 1)
    km = mask.flatten(1)
    for i in range(layers):
        x = encoder_layer(x, pe, km, sd, "%sencoder.layers.%d." % (p, i), training=training)
    memory = x
    t = torch.zeros_like(qe)
    for i in range(layers):
        t = decoder_layer(t, memory, pe, qe, km, sd, "%sdecoder.layers.%d." % (p, i), training=training)
    hs = layer_norm(t, sd, p + "decoder.norm.").transpose(0, 1)
    return hs, memory.permute(1, 2, 0).reshape(n, c, h, w)


def mlp(x, sd, p, num_layers):
    """detr.MLP: ReLU between layers, none after the last (reference detr.py:299-311)."""
    for i in range(num_layers):
        x = F.linear(x, sd["%slayers.%d.weight" % (p, i)], sd["%slayers.%d.bias" % (p, i)])
        if i < num_layers - 1:
            x = F.relu(x)
    return x


def detr_forward(sd, frames, mask, training=False):
    """DETR.forward (reference detr.py:48-75).  frames [n,3,H,W] f32, mask [n,H,W] (nonzero = padded)."""
    feat = resnet50_dc5(frames, sd)
    m = downsample_mask(mask, feat.shape[-2:])
    pos = sine_position(m)
    src = F.conv2d(feat, sd["input_proj.weight"], sd["input_proj.bias"])
    hs, memory = detr_transformer(src, m, sd["query_embed.weight"], pos, sd, training=training)
    return {
        "pred_logits": F.linear(hs, sd["class_embed.weight"], sd["class_embed.bias"]),
        "pred_boxes": mlp(hs, sd, "bbox_embed.", 3).sigmoid(),
        "image_features": feat,
        "embedded_memory_features": memory,
        "box_features": hs,
    }


# --------------------------------------------------------------------------------------
# parameter bookkeeping (reference utils/meta_utils.py:5-24 applied to the DETR module tree)
# --------------------------------------------------------------------------------------

def detr_state_shapes(num_classes=1235, num_queries=50, d=256, ffn=2048):
    """Ordered name -> shape of every DETR parameter and buffer, in ``state_dict()`` order."""
    out = {}

    def lin(p, o, i):
        out[p + "weight"] = (o, i)
        out[p + "bias"] = (o,)

    def ln(p):
        out[p + "weight"] = (d,)
        out[p + "bias"] = (d,)

    def attn(p):
        out[p + "in_proj_weight"] = (3 * d, d)
        out[p + "in_proj_bias"] = (3 * d,)
        lin(p + "out_proj.", d, d)

    for i in range(6):
        p = "transformer.encoder.layers.%d." % i
        attn(p + "self_attn.")
        lin(p + "linear1.", ffn, d)
        lin(p + "linear2.", d, ffn)
        ln(p + "norm1.")
        ln(p + "norm2.")
    for i in range(6):
        p = "transformer.decoder.layers.%d." % i
        attn(p + "self_attn.")
        attn(p + "multihead_attn.")
        lin(p + "linear1.", ffn, d)
        lin(p + "linear2.", d, ffn)
        ln(p + "norm1.")
        ln(p + "norm2.")
        ln(p + "norm3.")
    ln("transformer.decoder.norm.")
    lin("class_embed.", num_classes + 1, d)
    lin("bbox_embed.layers.0.", d, d)
    lin("bbox_embed.layers.1.", d, d)
    lin("bbox_embed.layers.2.", 4, d)
    out["query_embed.weight"] = (num_queries, d)
    out["input_proj.weight"] = (d, 2048, 1, 1)
    out["input_proj.bias"] = (d,)

    def bn(p, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            out[p + leaf] = (c,)

    b = "backbone.0.body."
    out[b + "conv1.weight"] = (64, 3, 7, 7)
    bn(b + "bn1.", 64)
    inplanes = 64
    for (name, planes, _s, _d0, _d), nblocks in zip(LAYER_CFG, RESNET50_BLOCKS):
        for k in range(nblocks):
            p = "%s%s.%d." % (b, name, k)
            out[p + "conv1.weight"] = (planes, inplanes, 1, 1)
            bn(p + "bn1.", planes)
            out[p + "conv2.weight"] = (planes, planes, 3, 3)
            bn(p + "bn2.", planes)
            out[p + "conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(p + "bn3.", planes * 4)
            if k == 0:
                out[p + "downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                bn(p + "downsample.1.", planes * 4)
            inplanes = planes * 4
    return out


def is_frozen(name):
    """conv1 / layer1 of the backbone never train (reference backbone.py:61-63); BN entries are buffers."""
    if not name.startswith("backbone."):
        return False
    if ".bn" in name or "downsample.1." in name:
        return True
    return not any(l in name for l in ("layer2", "layer3", "layer4"))


def theta_names(shapes=None):
    """Names of the tensors ``get_parameters(detector)`` returns, in its order.

    The helper only collects parameters of modules without children, so
    ``nn.MultiheadAttention`` contributes ``out_proj.*`` but not ``in_proj_*``
    (reference meta_utils.py:5-24).  state_dict order equals the
    recursive-children order, so a filter is enough.
    """
    shapes = shapes or detr_state_shapes()
    return [k for k in shapes if not is_frozen(k) and "in_proj_" not in k]


def trainable_names(shapes=None):
    shapes = shapes or detr_state_shapes()
    return [k for k in shapes if not is_frozen(k)]
