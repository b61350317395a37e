"""Autograd bindings of the HIP kernels for the training forward (reference train.py:175-181).

`conv_bn_act(conv, bn, x, act, residual)` runs Conv2d -> frozen BatchNorm2d -> (+ residual) -> ReLU as ONE launch of
fd_conv2d_nhwc_f32 (the BN is folded into the epilogue, as at inference) and differentiates it with
  * the ReLU mask applied to the incoming gradient (one elementwise pass, from the saved output),
  * the data gradient = the same conv kernel on dY with flipped / transposed weights (stride-1 layers),
  * the weight gradient = fd_conv2d_bwd_weight_f32 (any stride),
on channels-last tensors, which ARE the library's NHWC rows (zero-copy views).  Depthwise 3x3 layers use
fd_dwconv3x3_nhwc (forward and data gradient) and fd_dwconv3x3_bwd_weight_nhwc.  What the kernels do not cover falls
back to stock PyTorch-ROCm ops on the GPU: the data gradient of strided layers, dense layers with Cin % 32 != 0 or
Cout % 4 != 0 (the 7x7 stem when it is trainable, the 1-channel centre-ness conv), BatchNorm in training mode or with
trainable affine parameters, and SiLU (kept outside the fused epilogue because its derivative needs the pre-activation).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from ._lib import FdError, Segs
from .ops import ACT_NONE, ACT_RELU, Rows

_STOCK = os.environ.get("FD_TRAIN_STOCK_CONV") == "1"   # diagnostic: route every conv to the stock op (timing comparisons)
STATS = {"cl_copies": 0}                                 # activation-sized layout copies made on entry (should stay 0)


def _rows(t: torch.Tensor) -> Rows:
    """NCHW-shaped channels-last tensor -> [B*H*W, C] rows view (no copy)."""
    B, Cc, H, W = t.shape
    return Rows(t.permute(0, 2, 3, 1).reshape(B * H * W, Cc))


def _cl(t: torch.Tensor) -> torch.Tensor:
    if t.is_contiguous(memory_format=torch.channels_last):
        return t
    STATS["cl_copies"] += 1
    return t.contiguous(memory_format=torch.channels_last)


def _pad_of(m: nn.Conv2d) -> int:
    if isinstance(m.padding, str):
        return m.dilation[0] * (m.kernel_size[0] - 1) // 2
    return m.padding[0]


def _square(m: nn.Conv2d) -> bool:
    return (m.kernel_size[0] == m.kernel_size[1] and m.stride[0] == m.stride[1] and m.dilation[0] == m.dilation[1]
            and (isinstance(m.padding, str) or m.padding[0] == m.padding[1]) and m.padding_mode == "zeros")


def _dense_ok(m: nn.Conv2d, x: torch.Tensor) -> bool:
    return (m.groups == 1 and m.in_channels % 32 == 0 and m.out_channels % 4 == 0 and _square(m)
            and x.dtype == torch.float32 and m.weight.dtype == torch.float32)


def _dw_ok(m: nn.Conv2d, x: torch.Tensor) -> bool:
    c4 = m.in_channels // 4
    return (m.groups == m.in_channels == m.out_channels and m.in_channels % 4 == 0 and m.kernel_size == (3, 3)
            and m.stride == (1, 1) and m.dilation == (1, 1) and _pad_of(m) == 1 and m.padding_mode == "zeros"
            and ((c4 < 256 and 256 % c4 == 0) or c4 % 256 == 0) and x.dtype == torch.float32)


def bn_is_frozen(bn: Optional[nn.Module]) -> bool:
    return (isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.track_running_stats and bn.affine
            and not bn.weight.requires_grad and not bn.bias.requires_grad)


def _bn_fold(bn: nn.BatchNorm2d):
    """(scale, shift) of a frozen BatchNorm2d, cached on the module until any of its tensors is written to."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_mean.data_ptr())
    hit = getattr(bn, "_fd_fold", None)
    if hit is None or hit[0] != key:
        hit = (key, ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps))
        bn._fd_fold = hit
    return hit[1]


def _conv_launch(x: torch.Tensor, w_oihw: torch.Tensor, y: torch.Tensor, *, k, stride, pad, dil, scale=None, shift=None,
                 res: Optional[torch.Tensor] = None, act=ACT_NONE) -> None:
    """y = act(conv(x, w) * scale + shift + res) on channels-last NCHW-shaped tensors."""
    B, Cin, H, W = x.shape
    Cout = w_oihw.shape[0]
    segs = Segs.make(B, [(H, W)])
    out_rows = B * y.shape[2] * y.shape[3]
    code = ops.heuristic_conv(out_rows, Cout, (Cin // 32) * k * k, True)
    tile, ksplit = code & 0xFF, max(1, code >> 8)
    ws = None
    if ksplit > 1:
        nb = _lib.lib().fd_conv_workspace_bytes(out_rows, Cout, ksplit)
        ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=x.device)
    ops.conv_call(_rows(x), segs, ops.pack_conv_weight(w_oihw), _rows(y), Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad,
                  dil=dil, scale=scale, shift=shift, res=_rows(res) if res is not None else None, act=act, tile=tile,
                  ksplit=ksplit, workspace=ws)()


class _HipConv2d(torch.autograd.Function):
    """y = act(conv(x, w) * scale + shift + residual); scale is a constant (frozen BN), shift may carry a gradient."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, residual, stride, pad, dil, act):
        xc = _cl(x)
        B, Cin, H, W = xc.shape
        Cout, _, k, _ = weight.shape
        Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        y = torch.empty(B, Cout, Ho, Wo, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        rc = _cl(residual) if residual is not None else None
        _conv_launch(xc, weight.detach(), y, k=k, stride=stride, pad=pad, dil=dil, scale=scale,
                     shift=shift.detach().contiguous() if shift is not None else None, res=rc, act=act)
        ctx.save_for_backward(xc, weight, scale, y if act == ACT_RELU else None)
        ctx.geom = (stride, pad, dil, act)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, weight, scale, y = ctx.saved_tensors
        stride, pad, dil, act = ctx.geom
        g = _cl(gy)
        if act == ACT_RELU:
            g = torch.ops.aten.threshold_backward(g, y, 0.0)
        B, Cin, H, W = xc.shape
        Cout, _, k, _ = weight.shape
        gx = gw = gshift = gres = None
        if ctx.needs_input_grad[4]:
            gres = g
        if ctx.needs_input_grad[0]:
            weff = weight.detach() if scale is None else weight.detach() * scale.view(-1, 1, 1, 1)
            if stride == 1 and Cout % 32 == 0:
                gx = torch.empty_like(xc)
                _conv_launch(g, weff.flip(2, 3).transpose(0, 1), gx, k=k, stride=1, pad=dil * (k - 1) - pad, dil=dil)
            else:  # strided layers / narrow outputs: stock op for the data gradient
                gx = torch.ops.aten.convolution_backward(g, xc, weff, None, [stride, stride], [pad, pad], [dil, dil], False,
                                                         [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = ops.conv_wgrad(_rows(xc), _rows(g), Segs.make(B, [(H, W)]), Cin=Cin, Cout=Cout, k=k, stride=stride,
                                pad=pad, dil=dil)
            if scale is not None:
                dw = dw * scale.view(-1, 1, 1, 1)
            gw = dw.permute(0, 3, 1, 2)                      # OHWI -> OIHW view
        if ctx.needs_input_grad[3]:
            gshift = g.sum(dim=(0, 2, 3))
        return gx, gw, None, gshift, gres, None, None, None, None


class _HipDwConv3x3(torch.autograd.Function):
    """Depthwise 3x3 (stride 1, pad 1, no bias): y = act(dw(x, w) * scale + shift), scale / shift constants."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, act):
        xc = _cl(x)
        B, Cc, H, W = xc.shape
        y = torch.empty_like(xc)
        ops.dwconv3x3(_rows(xc), ops.pack_dw_weight(weight), _rows(y), Segs.make(B, [(H, W)]), scale, shift, act)
        ctx.save_for_backward(xc, weight, scale, y if act == ACT_RELU else None)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, weight, scale, y = ctx.saved_tensors
        g = _cl(gy)
        if ctx.act == ACT_RELU:
            g = torch.ops.aten.threshold_backward(g, y, 0.0)
        B, Cc, H, W = xc.shape
        segs = Segs.make(B, [(H, W)])
        gx = gw = None
        if ctx.needs_input_grad[0]:
            weff = weight.detach() if scale is None else weight.detach() * scale.view(-1, 1, 1, 1)
            gx = torch.empty_like(xc)
            ops.dwconv3x3(_rows(g), ops.pack_dw_weight(weff.flip(2, 3)), _rows(gx), segs)
        if ctx.needs_input_grad[1]:
            dw = ops.dwconv3x3_wgrad(_rows(xc), _rows(g), segs)             # [9][C]
            if scale is not None:
                dw = dw * scale
            gw = dw.t().reshape(Cc, 1, 3, 3)
        return gx, gw, None, None, None


def _need_cuda(x: torch.Tensor) -> None:
    if not x.is_cuda:
        raise FdError("pytorch_object_detection_amd runs on the GPU only; there is no CPU fallback (got a CPU tensor)")


def conv_bn_act(m: nn.Conv2d, bn: Optional[nn.Module], x: torch.Tensor, act: int = ACT_NONE,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(bn(m(x)) + residual), act in {ACT_NONE, ACT_RELU}; one HIP launch when the layer is covered (module docstring)."""
    _need_cuda(x)
    fold = bn is None or bn_is_frozen(bn)
    if not _STOCK and fold and (_dense_ok(m, x) or _dw_ok(m, x)):
        scale = shift = None
        if bn is not None:
            scale, shift = _bn_fold(bn)
        if m.groups == 1:
            if m.bias is not None:
                shift = m.bias if scale is None else m.bias * scale + shift
            return _HipConv2d.apply(x, m.weight, scale, shift, residual, m.stride[0], _pad_of(m), m.dilation[0], act)
        if m.bias is None and residual is None:
            return _HipDwConv3x3.apply(x, m.weight, scale, shift, act)
    if not _STOCK and bn is not None and not fold and (_dense_ok(m, x) or _dw_ok(m, x)):
        y = bn(conv_bn_act(m, None, x))                     # BN in training mode: conv on HIP, statistics stock
    else:
        y = m(x) if bn is None else bn(m(x))
    if residual is not None:
        y = y + residual
    return F.relu(y) if act == ACT_RELU else y


def conv2d(m: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """m(x) with the HIP kernels where they apply."""
    return conv_bn_act(m, None, x)


def stem_frozen(trunk: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """maxpool(relu(bn1(conv1(x)))) of a frozen ResNet stem on the inference kernels (no autograd graph)."""
    B, _, H, W = x.shape
    dev = x.device
    with torch.no_grad():
        x4 = torch.empty(B * H * W, 4, dtype=torch.float32, device=dev)
        ops.nchw3_to_nhwc4(x.contiguous(), x4)
        sc, sf = _bn_fold(trunk.bn1)
        s_in = Segs.make(B, [(H, W)])
        s1 = ops.conv_out_segs(s_in, 7, 2, 3, 1)
        H1, W1 = s1.H[0], s1.W[0]
        y1 = torch.empty(s1.rows, 64, dtype=torch.float32, device=dev)
        ops.conv_call(Rows(x4), s_in, ops.pack_stem_weight(trunk.conv1.weight), Rows(y1), Cin=4, Cout=64, k=7, stride=2,
                      pad=3, scale=sc, shift=sf, act=ACT_RELU, stem=True)()
        H2, W2 = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        y2 = torch.empty(B, 64, H2, W2, dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        ops.maxpool(Rows(y1), _rows(y2), B, H1, W1, 3, 2, 1)
    return y2


def stem_is_frozen(trunk: nn.Module, x: torch.Tensor) -> bool:
    return (not _STOCK and bn_is_frozen(trunk.bn1) and not trunk.conv1.weight.requires_grad and not x.requires_grad
            and x.dtype == torch.float32 and tuple(trunk.conv1.weight.shape) == (64, 3, 7, 7))
