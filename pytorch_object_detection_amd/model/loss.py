"""FCOS losses on MI355X — API of the reference's model/loss.py (compute_cls_loss :6-28, compute_cnt_loss :31-57,
compute_reg_loss :116-139, iou/giou :142-177, focal :180-193, FCOSLoss :196-215).  Each term is one fused masked HIP
forward kernel + one backward kernel (csrc/fd_loss.hip): no boolean-mask gathers, no per-image Python loop."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .._lib import FdError

_MODES = {"iou": 0, "giou": 1}


class _LtrbLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, mode):
        pred_c, target_c = pred.contiguous().float(), target.contiguous().float()
        mask_c = mask.contiguous().to(torch.uint8)
        loss, npos = ops.ltrb_loss_fwd(pred_c, target_c, mask_c, mode)
        ctx.save_for_backward(pred_c, target_c, mask_c, npos)
        ctx.mode = mode
        return loss / npos.clamp(min=1).float()

    @staticmethod
    def backward(ctx, g):
        pred, target, mask, npos = ctx.saved_tensors
        gscale = (g / npos.clamp(min=1).float()).contiguous()
        return ops.ltrb_loss_bwd(pred, target, mask, gscale, ctx.mode), None, None, None


def ltrb_reg_loss(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor, mode: str = "giou") -> torch.Tensor:
    """Per-image regression loss [B] = sum_{positives} loss(pred, target) / max(num_pos, 1)
    (compute_reg_loss, loss.py:116-139).  pred / target [B, L, 4] LTRB distances, mask [B, L] bool."""
    if mode not in _MODES:
        raise NotImplementedError("reg loss only implemented ['iou','giou']")
    return _LtrbLoss.apply(pred, target, mask, _MODES[mode])


class _FocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, num_pos):
        lg = logits.contiguous().float()
        lb = labels.contiguous().to(torch.int64)
        loss = ops.focal_loss_fwd(lg, lb)
        ctx.save_for_backward(lg, lb, num_pos)
        return loss / num_pos

    @staticmethod
    def backward(ctx, g):
        lg, lb, num_pos = ctx.saved_tensors
        return ops.focal_loss_bwd(lg, lb, (g / num_pos).contiguous()), None, None


class _BceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, mask):
        xc, tc = x.contiguous().float(), target.contiguous().float()
        mc = mask.contiguous().to(torch.uint8)
        loss, npos = ops.bce_loss_fwd(xc, tc, mc)
        ctx.save_for_backward(xc, tc, mc, npos)
        return loss / npos.clamp(min=1).float()

    @staticmethod
    def backward(ctx, g):
        xc, tc, mc, npos = ctx.saved_tensors
        return ops.bce_loss_bwd(xc, tc, mc, (g / npos.clamp(min=1).float()).contiguous()), None, None


def flatten_levels(preds) -> torch.Tensor:
    """list of NCHW maps -> [B, sum HW, C] (reshape_cat_out's layout, loss.py:124-128)."""
    return torch.cat([p.permute(0, 2, 3, 1).reshape(p.shape[0], -1, p.shape[1]) for p in preds], 1)


def compute_cls_loss(preds, target, mask) -> torch.Tensor:
    """Per-image focal loss / num_pos (loss.py:6-28).  target [B,L,1] int64 class ids (0 = background)."""
    num_pos = mask.sum(dim=1).clamp(min=1).float()
    return _FocalLoss.apply(flatten_levels(preds), target.squeeze(-1), num_pos)


def compute_cnt_loss(preds, target, mask) -> torch.Tensor:
    """Per-image BCE-with-logits over positives / num_pos (loss.py:31-57)."""
    return _BceLoss.apply(flatten_levels(preds).squeeze(-1), target.squeeze(-1), mask)


def compute_reg_loss(preds, target, mask, mode: str = 'iou') -> torch.Tensor:
    return ltrb_reg_loss(flatten_levels(preds), target, mask, mode)


class FCOSLoss(nn.Module):
    """FCOSLoss(mode)([preds, targets]) -> (cls_loss, cnt_loss, reg_loss, total) (loss.py:196-215): three fused HIP
    forward kernels and their backward kernels through torch.autograd.Function."""

    def __init__(self, mode: str = 'giou'):
        super().__init__()
        self.mode = mode

    def reg_loss(self, reg_preds, reg_target, cnt_target) -> torch.Tensor:
        mask_pos = (cnt_target > -1).squeeze(dim=-1)
        return compute_reg_loss(reg_preds, reg_target, mask_pos, self.mode).mean()

    def forward(self, x):
        pred, target = x
        cls_logit, cnt_logit, reg_logit = pred
        cls_target, cnt_target, reg_target = target
        mask_pos = (cnt_target > -1).squeeze(dim=-1)
        cls_loss = compute_cls_loss(cls_logit, cls_target, mask_pos).mean()
        cnt_loss = compute_cnt_loss(cnt_logit, cnt_target, mask_pos).mean()
        reg_loss = compute_reg_loss(reg_logit, reg_target, mask_pos, self.mode).mean()
        return cls_loss, cnt_loss, reg_loss, cls_loss + cnt_loss + reg_loss
