"""Layers used by HISFCOS (reference model/modules/modules.py:40-49,65-73,107-121,170-176) under the reference's names.
Inside a detector their arithmetic is fused into the HIP plan (engine.py) or the rows-based training forward; called on
their own -- the reference exposes them as ordinary layers -- `forward` runs the same HIP kernels (C-ABI), forward and
backward, and raises FdError for a configuration the kernels do not cover: there is no silent stock-op fallback."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..._lib import FdError


def _layer_conv(m: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """m(x) on the HIP kernels: dense convs (Cin % 32 == 0) and depthwise 3x3 stride 1 with autograd (train_ops nodes);
    other depthwise shapes (k in {3,5,7}, stride 1 / 2, padding k//2) without a backward."""
    from ... import ops, train_ops as T
    T._need_cuda(x)
    if T._STOCK:
        return nn.Conv2d.forward(m, x)           # FD_TRAIN_STOCK_CONV=1: the explicit stock-op diagnostic mode
    if (m.groups == 1 and T._dense_ok(m, x)) or T._dw_ok(m, x):
        return T.conv_bn_act(m, None, x)
    k, s = m.kernel_size[0], m.stride[0]
    if (m.groups == m.in_channels == m.out_channels and m.in_channels % 4 == 0 and T._square(m) and k in (3, 5, 7) and s in (1, 2)
            and m.dilation == (1, 1) and T._pad_of(m) == k // 2 and x.dtype == torch.float32):
        if torch.is_grad_enabled() and (x.requires_grad or m.weight.requires_grad):
            raise FdError("this depthwise shape has a HIP forward only (backward: 3x3 stride 1); run it under torch.no_grad()")
        B, C, H, W = x.shape
        p = k // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        xr = T.to_rows(x).contiguous()
        y = torch.empty(B * Ho * Wo, C, dtype=torch.float32, device=x.device)
        ops.dwconv2d(ops.Rows(xr), ops.pack_dwk_weight(m.weight), ops.Rows(y), B, H, W, k, s, p, p, Ho, Wo,
                     None, m.bias.detach() if m.bias is not None else None)
        return T.from_rows(y, B, Ho, Wo)
    raise FdError(f"{type(m).__name__}{tuple(m.weight.shape)}: not covered by the HIP kernels (dense: Cin % 32 == 0, Cout % 4 == 0, "
                  "square kernel, fp32; depthwise: C % 4 == 0, k in {3,5,7}, stride 1 / 2, padding k//2)")


class DepthWiseConv2d(nn.Conv2d):
    def __init__(self, in_channel: int, kernel: int, st: int = 1, bs: bool = False):
        super().__init__(in_channel, in_channel, kernel, st, kernel // 2, groups=in_channel, bias=bs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _layer_conv(self, x)


class PointWiseConv(nn.Conv2d):
    def __init__(self, in_channel: int, out_channel: int, kernel: int = 1, st: int = 1, bs: bool = False):
        super().__init__(in_channel, out_channel, kernel, st, kernel // 2, bias=bs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _layer_conv(self, x)


class SEBlock(nn.Module):
    """x * sigmoid(W2 silu(W1 mean_hw(x) + b1) + b2); excitation.{0,2} are the two 1x1 convs (fd_se_scale_nhwc / _bwd)."""

    def __init__(self, n_in: int, r: int = 4):
        super().__init__()
        self.squeeze = nn.AdaptiveAvgPool2d(1)
        self.excitation = nn.Sequential(nn.Conv2d(n_in, n_in // r, 1), nn.SiLU(), nn.Conv2d(n_in // r, n_in, 1),
                                        nn.Sigmoid())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ... import train_ops as T
        T._need_cuda(x)
        if not T._se_ok(self, x) and not T._STOCK:
            raise FdError("SEBlock: not covered by the HIP kernels (C % 4 == 0, C <= 4096, C/r <= 1024, fp32)")
        if T._STOCK:
            return x * self.excitation(self.squeeze(x))
        B, _, H, W = x.shape
        return T.from_rows(T.se_rows(self, T.to_rows(x), B, H * W), B, H, W)


class ScaleExp(nn.Module):
    """exp(x * scale): fused into the reg_pred conv epilogue (FD_ACT_EXP) inside the detector plans; on its own an
    elementwise HIP launch (fd_act_nhwc) when no gradient is needed, plain tensor ops when one is (the scale is trainable)."""

    def __init__(self, init_value: float = 1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([init_value], dtype=torch.float32))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ... import ops, train_ops as T
        T._need_cuda(x)
        if torch.is_grad_enabled() and (x.requires_grad or self.scale.requires_grad) or x.dtype != torch.float32 or x.shape[1] % 4:
            return torch.exp(x * self.scale)
        B, C, H, W = x.shape
        xr = T.to_rows(x).contiguous()
        y = torch.empty_like(xr)
        ops.act(ops.Rows(xr), ops.Rows(y), ops.ACT_EXP, float(self.scale.detach()))
        return T.from_rows(y, B, H, W)


class MNBlock(nn.Module):
    """x + PW2(SiLU(PW1(BN(dilated depthwise k x k (x))))) -- reference model/modules/modules.py:195-216, the block MNFCOS is built from
    (model/od/MNFcos.py:222-297).  As shipped the reference pads the depthwise conv with `dilated`, which keeps the map size only for
    k = 3, so its own residual add raises for the k = 5 / 7 blocks; here the padding is 'same' (dilated * (kernel - 1) / 2): identical
    for k = 3, the evident intent for 5 / 7.  Inside a detector plan the block is three HIP launches (engine.add_mn_block); called on
    its own (eval, no gradient) it runs the same launches."""

    def __init__(self, in_ch: int, out_ch: int, kernel: int, dilated: int, alpha: int = 1):
        super().__init__()
        self.DilatedDepthWiseConv = nn.Conv2d(in_ch, in_ch, kernel, 1, dilated * (kernel - 1) // 2, dilated, in_ch, False)
        self.BN = nn.BatchNorm2d(in_ch)
        self.PW1 = nn.Conv2d(in_ch, in_ch * alpha, 1, 1, 0, 1, 1, True)
        self.ACT1 = nn.SiLU(True)
        self.PW2 = nn.Conv2d(in_ch * alpha, out_ch, 1, 1, 0, 1, 1, True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ... import engine, train_ops as T
        from ..._lib import Segs
        T._need_cuda(x)
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or self.PW1.weight.requires_grad)):
            raise FdError("MNBlock has a HIP forward only (MNFCOS is inference-only on the HIP path): call .eval() and run it under torch.no_grad()")
        if x.dtype != torch.float32 or x.shape[1] % 4 or self.PW2.weight.shape[0] != x.shape[1]:
            raise FdError("MNBlock: fp32 input with C % 4 == 0 and out_ch == in_ch (the residual add) expected")
        B, C, H, W = x.shape
        plan = engine.Plan(x.device)
        segs = Segs.make(B, [(H, W)])
        xin, out = plan.pool.get(B * H * W, C), plan.pool.get(B * H * W, C)
        engine.add_mn_block(plan, "MNBlock", self, xin, segs, out)
        xin.tensor().view(B, H, W, C).copy_(x.permute(0, 2, 3, 1))
        plan.run()
        return out.tensor().view(B, H, W, C).permute(0, 3, 1, 2).clone()


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
