"""Layer containers used by HISFCOS (reference model/modules/modules.py:40-49,65-73,107-121,170-176).
They hold parameters under the reference's names; the arithmetic is fused into the HIP plan (engine.py)."""
from __future__ import annotations

import torch
import torch.nn as nn


class DepthWiseConv2d(nn.Conv2d):
    def __init__(self, in_channel: int, kernel: int, st: int = 1, bs: bool = False):
        super().__init__(in_channel, in_channel, kernel, st, kernel // 2, groups=in_channel, bias=bs)


class PointWiseConv(nn.Conv2d):
    def __init__(self, in_channel: int, out_channel: int, kernel: int = 1, st: int = 1, bs: bool = False):
        super().__init__(in_channel, out_channel, kernel, st, kernel // 2, bias=bs)


class SEBlock(nn.Module):
    """excitation.{0,2} are the two 1x1 convs; runs as fd_se_scale_nhwc."""

    def __init__(self, n_in: int, r: int = 4):
        super().__init__()
        self.squeeze = nn.AdaptiveAvgPool2d(1)
        self.excitation = nn.Sequential(nn.Conv2d(n_in, n_in // r, 1), nn.SiLU(), nn.Conv2d(n_in // r, n_in, 1),
                                        nn.Sigmoid())


class ScaleExp(nn.Module):
    """exp(x * scale): fused into the reg_pred conv epilogue (FD_ACT_EXP)."""

    def __init__(self, init_value: float = 1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([init_value], dtype=torch.float32))


def init_conv_random_normal(module: nn.Module, std: float = 0.01):
    if isinstance(module, nn.Conv2d):
        nn.init.normal_(module.weight, std=std)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)


def init_conv_kaiming(module: nn.Module):
    if isinstance(module, nn.Conv2d):
        nn.init.kaiming_uniform_(module.weight, a=1)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)
