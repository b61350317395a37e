"""FCOS losses on MI355X — API of the reference's model/loss.py.  The LTRB IoU / GIoU regression loss
(loss.py:116-177) is one fused masked HIP forward kernel + one backward kernel (csrc/fd_loss.hip)."""
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


def flatten_levels(preds) -> torch.Tensor:
    """list of NCHW maps -> [B, sum HW, C] (reshape_cat_out's layout, loss.py:124-128); a view-free gather."""
    return torch.cat([p.permute(0, 2, 3, 1).reshape(p.shape[0], -1, p.shape[1]) for p in preds], 1)


def compute_reg_loss(preds, target, mask, mode: str = 'iou') -> torch.Tensor:
    return ltrb_reg_loss(flatten_levels(preds), target, mask, mode)


class FCOSLoss(nn.Module):
    """FCOSLoss(mode)([preds, targets]) -> (cls, cnt, reg, total) (loss.py:196-215).  Only the regression term is
    a HIP kernel so far; the focal / BCE terms are SURVEY §8(f) 'next' rows and raise until built."""

    def __init__(self, mode: str = 'giou'):
        super().__init__()
        self.mode = mode

    def reg_loss(self, reg_preds, reg_target, cnt_target) -> torch.Tensor:
        mask_pos = (cnt_target > -1).squeeze(dim=-1)
        return compute_reg_loss(reg_preds, reg_target, mask_pos, self.mode).mean()

    def forward(self, x):
        raise FdError("FCOSLoss.forward: focal / centerness terms are not built yet (SURVEY.md §8f n2); "
                      "use FCOSLoss.reg_loss / ltrb_reg_loss for the HIP IoU/GIoU term")
