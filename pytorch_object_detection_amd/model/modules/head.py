"""Detection post-processing on MI355X — the reference's model/modules/head.py API (FCOSHead :41-102,
ClipBoxes :152-162) over the HIP decode -> top-k -> batched-NMS pipeline (csrc/fd_postproc.hip)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn as nn

from ... import ops
from ..._lib import FdError, Segs
from ...ops import Rows


def _as_pyramid(levels, nlev: int) -> Tuple[Rows, Segs]:
    """First `nlev` levels as (rows buffer, level table).  Zero-copy when `levels` is the model's PyramidOut."""
    rows, segs = getattr(levels, "rows", None), getattr(levels, "segs", None)
    if rows is not None and segs is not None:
        if nlev == segs.nseg:
            return rows, segs
        sub = Segs.make(segs.batch, segs.level_hw()[:nlev])  # zip(inputs, strides) drops trailing levels (head.py:20)
        return rows, sub
    # generic path: caller-owned NCHW tensors -> one NHWC rows buffer (layout plumbing only)
    ts = list(levels)[:nlev]
    for t in ts:
        if not t.is_cuda:
            raise FdError("FCOSHead runs on the GPU only; there is no CPU fallback (got a CPU tensor)")
    B, C = ts[0].shape[0], ts[0].shape[1]
    segs = Segs.make(B, [(t.shape[2], t.shape[3]) for t in ts])
    cpad = C if C % 4 == 0 else C + 4 - C % 4
    buf = torch.empty(segs.rows, cpad, dtype=torch.float32, device=ts[0].device)
    for i, t in enumerate(ts):
        buf[segs.m_start[i]:segs.m_start[i + 1], :C].view(B, t.shape[2], t.shape[3], C).copy_(t.permute(0, 2, 3, 1))
    return Rows(buf, 0, C), segs


class FCOSHead(nn.Module):
    """FCOSHead(score_threshold, nms_threshold, max_detection_box, strides)(model_out) -> (scores, classes, boxes).

    Semantics of head.py:52-102: score = sqrt(max_c sigmoid(cls) * sigmoid(cnt)), class = argmax + 1, LTRB decode
    around (x*s + s//2, y*s + s//2), per-image top-k (K = min(max_box, sum HW)), score >= threshold, per-class NMS
    through the class-offset trick of torchvision.ops.batched_nms, results score-descending.
    `detect_padded` returns the ragged result as padded [B, K] tensors + counts without any host sync;
    `forward` reproduces the reference's stacked return (and its failure when images keep different counts).
    """

    def __init__(self, score_threshold: float, nms_threshold: float, max_detection_box: int, strides: List[int]):
        super().__init__()
        self.score = score_threshold
        self.nms_threshold = nms_threshold
        self.max_box = max_detection_box
        self.strides = strides

    def decode_topk(self, x):
        """Stage head.py:52-82: (scores [B,K], classes [B,K] int64, boxes [B,K,4]) = the input of post_process."""
        nlev = min(len(self.strides), len(x[0]))
        cls, segs = _as_pyramid(x[0], nlev)
        cnt, _ = _as_pyramid(x[1], nlev)
        reg, _ = _as_pyramid(x[2], nlev)
        scores, classes, boxes = ops.fcos_decode(cls, cnt, reg, segs, list(self.strides)[:nlev])
        k = min(self.max_box, scores.shape[1])
        return ops.fcos_topk(scores, classes, boxes, k)

    def detect_padded(self, x):
        """-> scores [B,K], classes [B,K] int64, boxes [B,K,4], counts [B] int32; rows >= counts[b] are zero."""
        s, c, b = self.decode_topk(x)
        os_, oc, ob, _, counts = ops.batched_nms(s, c, b, float(self.score), float(self.nms_threshold))
        return os_, oc, ob, counts

    def post_process(self, preds_top_k):
        """head.py:84-102 on arbitrary (scores [B,K], classes [B,K], boxes [B,K,4]): the reference masks any position and
        torchvision's nms sorts internally, while fd_batched_nms wants score-descending rows (what decode_topk produces) --
        so rows are put in descending score order first (stable: ties keep their order); NaN scores are rejected."""
        s, c, b = preds_top_k
        if s.shape[1] > 1 and not bool((s[:, 1:] <= s[:, :-1]).all()):         # also False when a NaN is present
            if bool(torch.isnan(s).any()):
                raise FdError("FCOSHead.post_process: NaN scores")
            order = torch.sort(s, dim=1, descending=True, stable=True)[1]
            s = torch.gather(s, 1, order)
            c = torch.gather(c, 1, order)
            b = torch.gather(b, 1, order[..., None].expand(-1, -1, 4))
        os_, oc, ob, _, counts = ops.batched_nms(s.contiguous(), c.contiguous().to(torch.int64), b.contiguous(), float(self.score),
                                                 float(self.nms_threshold))
        return self._stack(os_, oc, ob, counts)

    @staticmethod
    def _stack(os_, oc, ob, counts):
        cnt = counts.tolist()  # the only device->host sync: the result shape depends on it (as in the reference)
        if any(c != cnt[0] for c in cnt):
            # the reference torch.stack()s per-image results (head.py:99-101) and fails the same way
            raise RuntimeError(f"stack expects each tensor to be equal size, but images kept {cnt} boxes; "
                               "use FCOSHead.detect_padded() for ragged batches")
        k = cnt[0]
        return os_[:, :k], oc[:, :k], ob[:, :k]

    def forward(self, x):
        return self._stack(*self.detect_padded(x))


class ClipBoxes(nn.Module):
    """ClipBoxes()(batch_imgs, batch_boxes): clamp to [0, W-1] x [0, H-1] in place (head.py:152-162)."""

    def __init__(self):
        super().__init__()

    @staticmethod
    def forward(batch_imgs: torch.Tensor, batch_boxes: torch.Tensor) -> torch.Tensor:
        h, w = batch_imgs.shape[2:]
        if not batch_boxes.is_contiguous():
            clipped = ops.clip_boxes_(batch_boxes.contiguous(), h, w)
            batch_boxes.copy_(clipped)
            return batch_boxes
        return ops.clip_boxes_(batch_boxes, h, w)


class FCOSGenTargets(nn.Module):
    """FCOSGenTargets(strides, limit_range)([out, gt_boxes, classes]) -> (cls_target [B,L,1] int64,
    cnt_target [B,L,1], reg_target [B,L,4]) (head.py:211-316) as one HIP kernel (fd_fcos_gen_targets): one thread
    per (image, location) loops over the GT boxes in registers instead of materialising [B,HW,M,4] temporaries."""

    def __init__(self, strides: List[int], limit_range: List[List[int]]):
        super().__init__()
        assert len(strides) == len(limit_range)
        self.stride, self.lim_range = strides, limit_range

    def forward(self, x):
        cls_logit = x[0][0]
        gt_box, labels = x[1], x[2]
        assert len(self.stride) == len(cls_logit)
        level_hw = [(int(t.shape[2]), int(t.shape[3])) for t in cls_logit]
        return ops.fcos_gen_targets(gt_box, labels, level_hw, self.stride, self.lim_range, 1.5)
