"""Minimal stand-in for the absent ``torchvision`` package, used ONLY by
``make_golden.py`` to import the reference in the build container.

It restates the public torchvision contract the reference touches:
  * ``torchvision.models.resnet50(replace_stride_with_dilation, pretrained, norm_layer)``
    -- ResNet-50 v1.5 (stride on the 3x3 conv), state_dict keys
    ``conv1 / bn1 / layerX.Y.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.0,downsample.1} / fc``
  * ``torchvision.models._utils.IntermediateLayerGetter``
  * ``torchvision.ops.boxes.box_area``, ``torchvision.ops.{nms, box_iou}``,
    ``torchvision.ops.misc.interpolate``
  * ``torchvision.transforms`` / ``torchvision.transforms.functional`` (import-time names only)
It never travels to the GPU box as part of the product; tests do not import it.
"""
import sys
import types
from collections import OrderedDict

import torch
from torch import nn


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, dilation, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


class _ResNet50(nn.Module):
    def __init__(self, replace_stride_with_dilation, norm_layer):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, 3, 1, False, norm_layer)
        self.layer2 = self._make(128, 4, 2, replace_stride_with_dilation[0], norm_layer)
        self.layer3 = self._make(256, 6, 2, replace_stride_with_dilation[1], norm_layer)
        self.layer4 = self._make(512, 3, 2, replace_stride_with_dilation[2], norm_layer)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)

    def _make(self, planes, blocks, stride, dilate, norm_layer):
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), norm_layer(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, prev, down, norm_layer)]
        self.inplanes = planes * 4
        layers += [_Bottleneck(self.inplanes, planes, 1, self.dilation, None, norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        remaining, layers = dict(return_layers), OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(a, b):
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (box_area(a)[:, None] + box_area(b) - inter)


def nms(boxes, scores, iou_threshold):
    order = scores.argsort(descending=True).tolist()
    keep = []
    while order:
        i = order.pop(0)
        keep.append(i)
        if order:
            ious = box_iou(boxes[i:i + 1], boxes[order])[0]
            order = [o for o, v in zip(order, ious.tolist()) if v <= iou_threshold]
    return torch.tensor(keep, dtype=torch.long)


def install():
    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.10.0"
    tv._is_tracing = lambda: False
    models = types.ModuleType("torchvision.models")
    models.resnet50 = lambda replace_stride_with_dilation=(False, False, False), pretrained=False, norm_layer=None: \
        _ResNet50(list(replace_stride_with_dilation), norm_layer or nn.BatchNorm2d)
    utils = types.ModuleType("torchvision.models._utils")
    utils.IntermediateLayerGetter = IntermediateLayerGetter
    models._utils = utils
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")
    boxes.box_area = box_area
    misc = types.ModuleType("torchvision.ops.misc")
    misc.interpolate = torch.nn.functional.interpolate
    ops.boxes, ops.misc, ops.nms, ops.box_iou = boxes, misc, nms, box_iou
    transforms = types.ModuleType("torchvision.transforms")
    functional = types.ModuleType("torchvision.transforms.functional")
    transforms.functional = functional

    class _Any:
        def __init__(self, *a, **k):
            pass

    for n in ("Compose", "Normalize", "ToPILImage", "ToTensor", "RandomCrop", "RandomErasing"):
        setattr(transforms, n, _Any)
    tv.models, tv.ops, tv.transforms = models, ops, transforms
    for name, mod in (("torchvision", tv), ("torchvision.models", models), ("torchvision.models._utils", utils),
                      ("torchvision.ops", ops), ("torchvision.ops.boxes", boxes), ("torchvision.ops.misc", misc),
                      ("torchvision.transforms", transforms), ("torchvision.transforms.functional", functional)):
        sys.modules[name] = mod
