"""DETR ResNet-50-DC5 detector on the HIP kernels, behind the reference's module tree / forward contract.

reference: models/detr_models/detr.py:21-83,299-341 (DETR, MLP, build), backbone.py:19-118, transformer.py:18-296,
position_encoding.py:12-48.  Attribute names and child order follow the reference so that ``state_dict()`` keys
(``backbone.0.body.layer2.0.conv1.weight`` ...) and the recursive-children parameter order used by the MAML
helpers are identical.

Internal layout is channels-last: activations are NHWC in the backbone and [frames, tokens, channels] in the
transformer; the NCHW tensors the reference returns are zero-copy permuted views.
"""
import torch
from torch import nn

from . import hipops as ops
from .nn import (Conv2dNHWC, Dropout, Embedding, FrozenBatchNorm2d, LayerNorm, Linear, MultiheadAttention,
                 PointwiseConv2d)


class NestedTensor(object):
    """reference models/detr_models/util/misc.py:282-302."""

    def __init__(self, tensors, mask):
        self.tensors = tensors
        self.mask = mask
        self.stem = None   # optional: precomputed frozen stem features of `tensors` (ResNet50Body.frozen_stem)

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


class Bottleneck(nn.Module):
    """torchvision ResNet v1.5 bottleneck (stride on the 3x3)."""

    def __init__(self, inplanes, planes, stride, dilation, downsample):
        super().__init__()
        self.conv1 = Conv2dNHWC(inplanes, planes, 1)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = Conv2dNHWC(planes, planes, 3, stride, dilation, dilation)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = Conv2dNHWC(planes, planes * 4, 1)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x, two_consumers=False):
        """x: the block input, or the PAIR of aliases the previous block handed out (two_consumers=True there): the block input
        feeds conv1 and the identity branch, and the sum of their two gradients then happens inside the previous tail's ReLU
        derivative (hipops.ReluBwdSum) instead of in a pass of its own (hipops.Fanout: the first block after the frozen stem)."""
        x, idt = x if isinstance(x, tuple) else ops.fanout(x, 2)
        if self.downsample is not None:
            idt = _conv_bn(self.downsample[0], self.downsample[1], idt)
        y = _conv_bn(self.conv1, self.bn1, x, relu=True)
        y = _conv_bn(self.conv2, self.bn2, y, relu=True)
        return _conv_bn(self.conv3, self.bn3, y, residual=idt, relu=True, fan=2 if two_consumers else 1)


def _conv_bn(conv, bn, x, residual=None, relu=False, fan=1):
    """bn(conv(x)) (+ residual) (+ ReLU) with the frozen-BN affine riding on the convolution's contraction (hipops.conv2d_nhwc_bn_act)"""
    scale, shift = bn.folded()
    return ops.conv2d_nhwc_bn_act(x, conv.weight, scale, shift, residual, relu, conv.stride, conv.padding, conv.dilation, fan)


def _make_layer(inplanes, planes, blocks, stride, first_dilation, dilation):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(Conv2dNHWC(inplanes, planes * 4, 1, stride), FrozenBatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, first_dilation, down)]
    layers += [Bottleneck(planes * 4, planes, 1, dilation, None) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class ResNet50Body(nn.Module):
    """conv1..layer4 of resnet50(replace_stride_with_dilation=[False, False, True]) (reference backbone.py:88-90)."""

    def __init__(self):
        super().__init__()
        self.conv1 = Conv2dNHWC(3, 64, 7, 2, 3)
        self.bn1 = FrozenBatchNorm2d(64)
        self.layer1 = _make_layer(64, 64, 3, 1, 1, 1)
        self.layer2 = _make_layer(256, 128, 4, 2, 1, 1)
        self.layer3 = _make_layer(512, 256, 6, 2, 1, 1)
        self.layer4 = _make_layer(1024, 512, 3, 1, 1, 2)

    def frozen_stem(self, frames_nchw):
        """conv1 / bn1 / maxpool / layer1: frozen and fed by inputs without gradient (reference backbone.py:61-63), so the
        result depends on the frames only -- the episode models compute it once per chunk and reuse it for the adapted
        forwards (``NestedTensor.stem``)."""
        n, c, H, W = frames_nchw.shape
        with torch.no_grad():
            g = ops.conv_geom(n, H, W, c, 7, 7, 2, 3, 1)
            cols = ops.im2col_any_layout(frames_nchw, g, channels_last=False)
            w = self.conv1.weight.reshape(64, -1)   # stored [out, kh, kw, in]
            if g.Kp != w.shape[1]:
                w = torch.nn.functional.pad(w, (0, g.Kp - w.shape[1]))
            x = ops.linear(cols, w).reshape(n, g.OH, g.OW, 64)
            x = self.bn1(x, relu=True)
            x = ops.maxpool_nhwc(x, 3, 2, 1)
            if ops.b16_active():   # MODEL.COMPUTE_DTYPE bf16: from here on activations live in HBM as bf16 (b16.py)
                from . import b16
                if b16.STEM_LAYER1:   # (layer1 -- frozen, 64 / 256 channels at 1/4 resolution -- on the bf16 kernels as well)
                    return self.layer1(b16.cast_b16(x))
                return b16.cast_b16(self.layer1(x))
            return self.layer1(x)

    def forward(self, frames_nchw, stem=None):
        x = stem if stem is not None else self.frozen_stem(frames_nchw)
        blocks = list(self.layer2) + list(self.layer3) + list(self.layer4)
        for i, blk in enumerate(blocks):   # (every tail but the last hands its output to the next block's two branches)
            x = blk(x, two_consumers=i + 1 < len(blocks))
        return x


class Backbone(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = ResNet50Body()
        self.num_channels = 2048
        for name, p in self.body.named_parameters():
            if "layer2" not in name and "layer3" not in name and "layer4" not in name:
                p.requires_grad_(False)

    def forward(self, tensor_list):
        feat = self.body(tensor_list.tensors, getattr(tensor_list, "stem", None))   # [n, h, w, 2048]
        m = tensor_list.mask
        assert m is not None
        mask = ops.mask_nearest((m != 0).to(torch.uint8).contiguous(), feat.shape[1], feat.shape[2])
        return feat, mask


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=128, temperature=10000.0):
        super().__init__()
        self.num_pos_feats, self.temperature = num_pos_feats, temperature

    def forward(self, mask_u8):
        return ops.sine_position(mask_u8, self.num_pos_feats, self.temperature)


class Joiner(nn.Module):
    """Children named "0" (backbone) and "1" (position embedding) like the reference's nn.Sequential."""

    def __init__(self, backbone, position_embedding):
        super().__init__()
        self.add_module("0", backbone)
        self.add_module("1", position_embedding)
        self.num_channels = backbone.num_channels

    def __getitem__(self, i):
        return getattr(self, str(i))

    def forward(self, tensor_list):
        feat, mask = self[0](tensor_list)
        return feat, mask, self[1](mask)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.dropout1 = Dropout(dropout)
        self.dropout2 = Dropout(dropout)

    def forward(self, src, key_padding_mask, pos):
        s_qk, s_v, s_res = ops.fanout(src, 3)
        qk = ops.add(s_qk, pos)
        a = self.self_attn(qk, qk, s_v, key_padding_mask, qk_same=True)
        s_ffn, s_res = ops.fanout(self.norm1(ops.add_dropout(s_res, a, self.dropout1.p, self.dropout1.training)), 2)
        f = self.linear2(ops.relu_dropout(self.linear1(s_ffn), self.dropout.p, self.dropout.training))
        return self.norm2(ops.add_dropout(s_res, f, self.dropout2.p, self.dropout2.training))


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.norm3 = LayerNorm(d_model)
        self.dropout1 = Dropout(dropout)
        self.dropout2 = Dropout(dropout)
        self.dropout3 = Dropout(dropout)

    def forward(self, tgt, memory, memory_key, memory_key_padding_mask, query_pos):
        """tgt [n,Q,E]; memory_key = memory + pos (shared by all layers); query_pos [Q*E] broadcast over frames, or
        [episodes, Q*E] (one learned query table per episode's fast weights, frames grouped by episode)."""
        n, Q, E = tgt.shape
        qp1, qp2 = query_pos if isinstance(query_pos, tuple) else (query_pos, query_pos)
        groups = qp1.shape[0] if qp1.dim() == 2 else 1
        t_qk, t_v, t_res = ops.fanout(tgt, 3)
        qk = ops.AddRowVec.apply(t_qk.reshape(n, Q * E), qp1, groups).reshape(n, Q, E)
        a = self.self_attn(qk, qk, t_v, None, qk_same=True)
        t_q, t_res = ops.fanout(self.norm1(ops.add_dropout(t_res, a, self.dropout1.p, self.dropout1.training)), 2)
        q = ops.AddRowVec.apply(t_q.reshape(n, Q * E), qp2, groups).reshape(n, Q, E)
        c = self.multihead_attn(q, memory_key, memory, memory_key_padding_mask)
        t_ffn, t_res = ops.fanout(self.norm2(ops.add_dropout(t_res, c, self.dropout2.p, self.dropout2.training)), 2)
        f = self.linear2(ops.relu_dropout(self.linear1(t_ffn), self.dropout.p, self.dropout.training))
        return self.norm3(ops.add_dropout(t_res, f, self.dropout3.p, self.dropout3.training))


class TransformerEncoder(nn.Module):
    def __init__(self, d_model, nhead, ffn, dropout, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(TransformerEncoderLayer(d_model, nhead, ffn, dropout) for _ in range(num_layers))
        self.num_layers = num_layers
        self.norm = None

    def forward(self, src, key_padding_mask, pos):
        for layer in self.layers:
            src = layer(src, key_padding_mask, pos)
        return src


class TransformerDecoder(nn.Module):
    def __init__(self, d_model, nhead, ffn, dropout, num_layers, norm=True):
        super().__init__()
        self.layers = nn.ModuleList(TransformerDecoderLayer(d_model, nhead, ffn, dropout) for _ in range(num_layers))
        self.num_layers = num_layers
        self.norm = LayerNorm(d_model) if norm else None

    def forward(self, tgt, memory, memory_key_padding_mask, pos, query_pos):
        L = len(self.layers)
        # the memory feeds every layer's cross-attention (as value, and + pos as key), the learned queries two adds per layer:
        # one gradient sum each at the end of the backward instead of a chain of two-operand adds (hipops.Fanout)
        if pos is None:
            mems = ops.fanout(memory, 2 * L)
            keys, mems = mems[:L], mems[L:]
        else:
            mems = ops.fanout(memory, L + 1)
            if pos.shape[0] == 1 and memory.shape[0] > 1:   # one fixed position table shared by a batch of sequences
                b = memory.shape[0]
                memory_key = ops.AddRowVec.apply(mems[L].reshape(b, -1), pos.reshape(-1), 1).reshape(memory.shape)
            else:
                memory_key = ops.add(mems[L], pos)
            keys = ops.fanout(memory_key, L)
        qps = ops.fanout(query_pos, 2 * L)
        for i, layer in enumerate(self.layers):
            tgt = layer(tgt, mems[i], keys[i], memory_key_padding_mask, (qps[2 * i], qps[2 * i + 1]))
        # return_intermediate=True in the reference, but only hs[-1] is consumed (detr.py:69): norm the last output
        return self.norm(tgt) if self.norm is not None else tgt


class Transformer(nn.Module):
    """reference transformer.py:18-58, on [frames, tokens, d] tensors."""

    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1):
        super().__init__()
        self.encoder = TransformerEncoder(d_model, nhead, dim_feedforward, dropout, num_encoder_layers)
        self.decoder = TransformerDecoder(d_model, nhead, dim_feedforward, dropout, num_decoder_layers)
        self.d_model, self.nhead = d_model, nhead

    def forward(self, src, mask, query_embed, pos):
        n, hw, E = src.shape
        Q = query_embed.shape[-2]
        memory, m_dec = ops.fanout(self.encoder(src, mask, pos), 2)   # (the caller's copy and the decoder's)
        tgt = torch.zeros(n, Q, E, device=src.device, dtype=src.dtype)
        qe = query_embed.reshape(Q * E) if query_embed.dim() == 2 else query_embed.reshape(-1, Q * E)
        hs = self.decoder(tgt, m_dec, mask, pos, qe)
        return hs, memory


class MLP(nn.Module):
    """reference detr.py:299-311."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(a, b) for a, b in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            # (16-bit mode: a head's last layer hands fp32 to the criterion / the decoders' consumers)
            x = layer(x) if i < self.num_layers - 1 else layer(x, out_dtype=torch.float32)
            if i < self.num_layers - 1:
                x = ops.Relu.apply(x)
        return x


class DETR(nn.Module):
    """reference detr.py:21-75.  forward(NestedTensor) -> the same five-entry dict."""

    def __init__(self, backbone, transformer, num_classes, num_queries):
        super().__init__()
        self.num_queries = num_queries
        self.transformer = transformer
        d = transformer.d_model
        self.class_embed = Linear(d, num_classes + 1)
        self.bbox_embed = MLP(d, d, 4, 3)
        self.query_embed = Embedding(num_queries, d)
        self.input_proj = PointwiseConv2d(backbone.num_channels, d)
        self.backbone = backbone
        self.aux_loss = False

    def forward(self, samples):
        frames = samples.tensors
        if not frames.is_cuda:
            raise RuntimeError("interactron_amd runs on the HIP kernels only: move the inputs to the GPU (no CPU path)")
        feat, mask, pos = self.backbone(samples)                    # [n,h,w,2048], [n,h,w] u8, [n,hw,256]
        n, h, w, c = feat.shape
        src = self.input_proj(feat.reshape(n, h * w, c))
        hs, memory = self.transformer(src, mask.reshape(n, h * w), self.query_embed.weight, pos)
        h_cls, h_box, hs = ops.fanout(hs, 3)
        return {
            "pred_logits": self.class_embed(h_cls, out_dtype=torch.float32),
            "pred_boxes": ops.Sigmoid.apply(self.bbox_embed(h_box)),
            "image_features": feat.permute(0, 3, 1, 2),
            "embedded_memory_features": memory.reshape(n, h, w, -1).permute(0, 3, 1, 2),
            "box_features": hs,
        }


def build_detector(num_classes, num_queries=50):
    backbone = Joiner(Backbone(), PositionEmbeddingSine(128))
    return DETR(backbone, Transformer(256, 8, 6, 6, 2048, 0.1), num_classes, num_queries)
