"""ResNet-50 v1.5 trunk containers with torchvision's parameter names.

The reference wraps torchvision.models.resnet50(pretrained=True) (model/backbone/resnet50.py:9-80).  torchvision is
not a dependency here and there is no network for pretrained weights: these classes only HOLD the parameters
(same state_dict keys, so the reference's checkpoints load) with torchvision's default initialisation; the
arithmetic runs in engine.build_resnet50 on the HIP conv kernel.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, downsample: bool):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)  # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))


def _make_layer(inplanes: int, planes: int, blocks: int, stride: int) -> nn.Sequential:
    layers = [_Bottleneck(inplanes, planes, stride, True)]
    layers += [_Bottleneck(planes * 4, planes, 1, False) for _ in range(blocks - 1)]
    return nn.Sequential(*layers)


class _Trunk(nn.Module):
    """conv1, bn1, layer1..layer4 (no avgpool/fc: the fx feature extractor of the reference prunes them)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = _make_layer(64, 64, 3, 1)
        self.layer2 = _make_layer(256, 128, 4, 2)
        self.layer3 = _make_layer(512, 256, 6, 2)
        self.layer4 = _make_layer(1024, 512, 3, 2)
        for m in self.modules():  # torchvision ResNet.__init__
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


def _no_torch_forward(self, *a, **k):
    raise RuntimeError("the backbone runs inside the model's HIP plan; call the detector (model(x)) instead")


def trunk_train_forward(trunk: nn.Module, x: torch.Tensor):
    """Autograd-capable trunk forward (training only; inference runs the HIP plan).  Every bottleneck conv is one fused
    HIP launch (conv + frozen BN + residual + ReLU, train_ops.conv_bn_act) differentiated by the HIP data- / weight-
    gradient kernels; a frozen stem (freeze_stages(1)) runs on the inference kernels.  Returns (C3, C4, C5)."""
    import torch.nn.functional as F

    from ...train_ops import bottleneck, stem_frozen, stem_is_frozen
    if stem_is_frozen(trunk, x):
        x = stem_frozen(trunk, x)
    else:                                                     # trainable 7x7 stem (Cin = 3): stock ops
        from ...train_ops import stock_fallback
        stock_fallback("a trainable / un-frozen 7x7 stem (Cin = 3)")
        x = F.max_pool2d(F.relu(trunk.bn1(trunk.conv1(x))), 3, 2, 1)
    feats = []
    for li in (1, 2, 3, 4):
        for blk in getattr(trunk, f"layer{li}"):
            x = bottleneck(blk, x)          # one autograd node per block (train_ops._BottleneckRows)
        feats.append(x)
    return feats[1], feats[2], feats[3]


class ResNet50v2(nn.Module):
    """Reference model/backbone/resnet50.py:59-97.  state_dict carries BOTH key sets the reference produces:
    backbone.{conv1,bn1,layer1}.* and backbone.extract_feature.{conv1,bn1,layer1..4}.* (shared parameters)."""

    def __init__(self):
        super().__init__()
        trunk = _Trunk()
        self.conv1 = trunk.conv1
        self.bn1 = trunk.bn1
        self.layer1 = trunk.layer1
        self.feature = ['layer2.3.relu_2', 'layer3.5.relu_2', 'layer4.2.relu_2']
        self.extract_feature = trunk

    @property
    def trunk(self) -> nn.Module:
        return self.extract_feature

    forward = _no_torch_forward

    def freeze_bn(self):
        for layer in self.modules():
            if isinstance(layer, (nn.BatchNorm2d, nn.SyncBatchNorm)):
                layer.eval()

    def freeze_stages(self, stage: int):
        if stage >= 0:
            self.bn1.eval()
            for m in [self.conv1, self.bn1]:
                for param in m.parameters():
                    param.requires_grad = False
        for i in range(1, stage + 1):
            layer = getattr(self, f'layer{i}')
            layer.eval()
            for param in layer.parameters():
                param.requires_grad = False


class ResNet50(nn.Module):
    """Reference model/backbone/resnet50.py:9-57 (FCOS baseline backbone): keys conv1, bn1, layer1..layer4."""

    def __init__(self, re_layer: int = 1):
        super().__init__()
        trunk = _Trunk()
        self.conv1, self.bn1 = trunk.conv1, trunk.bn1
        self.relu = nn.ReLU(inplace=True)
        self.max_pool = nn.MaxPool2d(3, 2, 1)
        self.layer1, self.layer2, self.layer3, self.layer4 = trunk.layer1, trunk.layer2, trunk.layer3, trunk.layer4
        self.re_layer = re_layer

    @property
    def trunk(self) -> nn.Module:
        return self

    forward = _no_torch_forward

    def freeze_bn(self):
        for layer in self.modules():
            if isinstance(layer, (nn.BatchNorm2d, nn.SyncBatchNorm)):
                layer.eval()

    def freeze_stages(self, stage: int):
        if stage >= 0:
            self.bn1.eval()
            for m in [self.conv1, self.bn1]:
                for param in m.parameters():
                    param.requires_grad = False
        for i in range(1, stage + 1):
            layer = getattr(self, 'layer{}'.format(i))
            layer.eval()
            for param in layer.parameters():
                param.requires_grad = False
