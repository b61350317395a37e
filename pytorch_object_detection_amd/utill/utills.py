"""Config loader + location grid with the reference's names (utill/utills.py:58-73, 258-272)."""
from __future__ import annotations

import os
from typing import List

import torch
from yaml import safe_load

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def coords_origin_fcos(feature: torch.Tensor, strides: int) -> torch.Tensor:
    """[H*W, 2] fp32 (x*s + s//2, y*s + s//2), x fastest, for an NHWC-shaped `feature` ([N, H, W, C]).
    The HIP decode kernel derives the same grid in-kernel; this helper exists for callers of the reference API."""
    h, w = feature.shape[1:3]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=feature.device) * strides,
                            torch.arange(w, dtype=torch.float32, device=feature.device) * strides, indexing='ij')
    return torch.stack([xs.reshape(-1), ys.reshape(-1)], -1) + strides // 2


def load_config(cfg: str = os.path.join(_PKG, 'config', 'main.yaml')) -> dict:
    """main.yaml names the dataset + model; the dataset yaml holds one block per model.  Result keys as in the
    reference: dataset_setting, <MODEL> blocks, model{dataset,name,amp,ddp,persistent,prefetch}, savename.
    Dataset yaml paths are resolved relative to main.yaml's directory's parent when not found as given."""
    with open(cfg) as f:
        main = safe_load(f)
    dataset = main['dataset']
    path = main[dataset]
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(cfg))), path)
    with open(path) as f:
        config = safe_load(f)
    config['model'] = {'dataset': dataset, 'name': main['model'], 'amp': main['amp'], 'ddp': main['ddp_enabled'],
                       'persistent': main['persistent_workers'], 'prefetch': main['prefetch_factor']}
    config['savename'] = main['savename']
    return config


class DataEncoder:
    """The two hot functions of the reference's DataEncoder (utill/utills.py:201-255) on the GPU.
    Anchor encode/decode (RetinaNet path) is outside the FCOS hot path and not provided."""

    def _box_iou(self, box1: torch.Tensor, box2: torch.Tensor, order: str = 'xyxy') -> torch.Tensor:
        """[N,4] x [M,4] -> [N,M] IoU with the '+1' pixel convention."""
        from .. import ops
        if order == 'xywh':          # utills.py:196-199 _change_box_order('xywh2xyxy'): (cx, cy, w, h) -> (c - wh/2, c + wh/2)
            box1 = torch.cat([box1[:, :2] - box1[:, 2:] / 2, box1[:, :2] + box1[:, 2:] / 2], 1)
            box2 = torch.cat([box2[:, :2] - box2[:, 2:] / 2, box2[:, :2] + box2[:, 2:] / 2], 1)
        elif order != 'xyxy':
            raise ValueError(f"unknown box order '{order}'")
        return ops.pairwise_iou(box1.contiguous().float(), box2.contiguous().float(), True)

    def _box_nms(self, bboxes: torch.Tensor, scores: torch.Tensor, threshold: float = 0.5, mode: str = 'union') -> torch.Tensor:
        """Greedy class-agnostic NMS, '+1' areas, keep while ovr <= threshold; returns kept indices (int64, score-descending)."""
        from .. import ops
        if mode not in ('union', 'min'):
            raise TypeError('Unknown nms mode: %s.' % mode)
        keep, counts = ops.box_nms_plus1(bboxes.contiguous().float()[None], scores.contiguous().float()[None], threshold, mode)
        return keep[0, :int(counts[0])].to(torch.int64)
