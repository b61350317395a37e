"""Shared plumbing of the detector modules: plan cache, input checks, pyramid-aware output lists."""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import torch
import torch.nn as nn

from ... import engine
from ..._lib import FdError, Segs
from ...ops import Rows


class PyramidOut(list):
    """List of per-level NCHW-shaped tensors (what the reference model returns) that also remembers the single
    NHWC rows buffer they are views of, so FCOSHead can consume it without any re-layout."""

    def __init__(self, views, rows: Rows, segs: Segs):
        super().__init__(views)
        self.rows, self.segs = rows, segs


def pyramid_out(rows: Rows, segs: Segs) -> PyramidOut:
    return PyramidOut(engine.level_views(rows, segs), rows, segs)


class PlannedModule(nn.Module):
    """nn.Module whose forward is a cached engine.Plan keyed by input shapes and the parameters' version counters."""

    def __init__(self):
        super().__init__()
        self._plans: Dict[Tuple, Tuple[int, object]] = {}
        # 'f32' (exact fp32 MFMA, default) | 'f16x3' (opt-in split-f16 products, ~1.6x faster, same 1e-4 parity bar);
        # None = the FD_CONV_PRECISION environment default.  Sub-modules inherit the value of the module that plans.
        self.conv_precision = None

    def _param_version(self) -> int:
        v = 0
        for t in list(self.parameters()) + list(self.buffers()):
            v += t._version + (t.data_ptr() & 0xFFFF)
        return v

    def _get_plan(self, key: Tuple, build: Callable[[], object]):
        ver = self._param_version()
        key = key + (self.conv_precision,)
        hit = self._plans.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        if len(self._plans) > 8:
            self._plans.clear()
        built = build()
        self._plans[key] = (ver, built)
        return built

    def invalidate_plans(self) -> None:
        self._plans.clear()

    max_plan_batch = None    # optional cap on the images one plan handles (larger batches run as consecutive sub-batches)

    def plan_batch_limit(self, x: torch.Tensor) -> int:
        """Images per plan: the conv kernels address each activation buffer through a 32-bit buffer descriptor
        (< 3 GiB); the largest map is the stem / layer1 output, H/2 * W/2 * 64 floats per image."""
        H, W = (x.shape[1], x.shape[2]) if x.dtype == torch.uint8 else (x.shape[2], x.shape[3])
        per_image = (H // 2) * (W // 2) * 64 * 4
        lim = max(1, int((3 * 2 ** 30 - 2 ** 24) // per_image))
        return min(lim, self.max_plan_batch) if self.max_plan_batch else lim

    def _forward_chunked(self, x: torch.Tensor, chunk: int):
        """Batches beyond plan_batch_limit: consecutive sub-batches through the (cached) plans, outputs copied out of the
        plan-owned buffers and concatenated per level (images are independent end to end)."""
        parts = []
        for i in range(0, x.shape[0], chunk):
            out = self.forward(x[i:i + chunk])
            parts.append([[t.clone() for t in grp] for grp in out])
        return tuple([torch.cat([p[g][lv] for p in parts], 0) for lv in range(5)] for g in range(3))


    @staticmethod
    def _check_image(x: torch.Tensor) -> None:
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3:
            raise FdError("expected an image batch [B, 3, H, W]")
        if not x.is_cuda:
            raise FdError("pytorch_object_detection_amd runs on the GPU only; there is no CPU fallback "
                          "(got a CPU tensor)")
        if x.dtype != torch.float32:
            raise FdError("expected float32 images (the HIP path computes in fp32)")
        if x.shape[2] % 32 or x.shape[3] % 32:
            raise FdError("H and W must be multiples of 32 (reference datasets pad to 32, dataset/voc.py:128-132)")

    def _check_eval(self) -> None:
        if self.training:
            raise FdError("the HIP plan implements the frozen-BN inference forward; call model.eval() first")

    def train(self, mode: bool = True):
        """nn.Module.train, except that BatchNorm layers the constructor froze stay in eval mode (the reference freezes
        them at construction, HISFcos.py:57-68, but its unguarded model.train() silently un-freezes the statistics;
        SURVEY.md §2 C3)."""
        super().train(mode)
        if mode and getattr(self, "backbone_freeze", False):
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    @staticmethod
    def _check_train_input(x: torch.Tensor) -> None:
        if not x.is_cuda:
            raise FdError("pytorch_object_detection_amd runs on the GPU only; there is no CPU fallback "
                          "(got a CPU tensor)")


def copy_in_nchw(dst: Rows, segs: Segs, level: int, x: torch.Tensor) -> None:
    """Stage a caller-owned NCHW tensor into level `level` of a plan-owned rows buffer (standalone sub-module calls)."""
    B, C, H, W = x.shape
    m0, m1 = segs.m_start[level], segs.m_start[level + 1]
    dst.tensor()[m0:m1].view(B, H, W, C).copy_(x.permute(0, 2, 3, 1))
