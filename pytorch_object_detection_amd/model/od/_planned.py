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
    _largest_map_channels = 64   # channels of the widest stride-2 activation (ResNet stem: 64)

    def plan_batch_limit(self, x: torch.Tensor) -> int:
        """Images per plan: the conv kernels address each activation buffer through a 32-bit buffer descriptor
        (< 3 GiB); the largest map is the stem / layer1 output, H/2 * W/2 * 64 floats per image (EfficientNet: the first
        stride-2 block's expanded input)."""
        H, W = (x.shape[1], x.shape[2]) if x.dtype == torch.uint8 else (x.shape[2], x.shape[3])
        per_image = (H // 2) * (W // 2) * self._largest_map_channels * 4
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

    pixel_mean = (0.485, 0.456, 0.406)   # dataset/voc.py:57-58
    pixel_std = (0.229, 0.224, 0.225)

    use_graph = False        # True: every plan's launches are captured into a HIP graph on its first forward and replayed afterwards
                             # (engine.Plan.capture_graph: the latency path -- batch 1 is launch-bound); results are identical
    copy_outputs = False     # True: forward returns fresh tensors (the reference's behaviour) instead of views of plan-owned
                             # buffers that the next forward of the same shape overwrites

    def outputs_of(self, plan):
        """(cls_logits, cnt_logits, reg_preds) of a plan after plan.run(): zero-copy channels-last views of the plan's output
        buffers (valid until the next forward of the same shape; FCOSHead consumes them in place), or -- copy_outputs --
        independent tensors a caller may keep across iterations."""
        outs = tuple(pyramid_out(o, plan.segs) for o in plan.outs)
        if self.copy_outputs:
            return tuple([t.clone() for t in grp] for grp in outs)
        return outs

    def plan_for(self, x, slot: int = 0):
        """The cached plan for an input: fp32 [B,3,H,W]; uint8 [B,H,W,3] (normalised on the device); or a list of resized
        uint8 [h, w, 3] images of different sizes (padded to the batch canvas + normalised on the device).  `slot` picks
        one of several independent plan instances of the same shape (own buffers: pipeline.TwoLanePipeline keeps two
        batches in flight)."""
        self._check_eval()
        if slot:
            real, self._get_plan = self._get_plan, (lambda key, build: real(key + ("slot", slot), build))
            try:
                return self.plan_for(x, 0)
            finally:
                del self._get_plan
        if isinstance(x, (list, tuple)):
            if not x or any((not isinstance(t, torch.Tensor)) or t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3 or not t.is_cuda
                            for t in x):
                raise FdError("forward_images expects a non-empty list of CUDA uint8 [h, w, 3] images")
            # dataset/voc.py:128-132: pad = 32 - n % 32 (a full extra 32 when already aligned); :149-156: batch maximum
            H = max(int(t.shape[0]) + 32 - int(t.shape[0]) % 32 for t in x)
            W = max(int(t.shape[1]) + 32 - int(t.shape[1]) % 32 for t in x)
            B, dev = len(x), x[0].device
            return self._get_plan(("model_collate", B, H, W, str(dev)), lambda: self.build_plan(B, H, W, dev, "collate"))
        if x.dtype == torch.uint8:       # [B, H, W, 3] uint8: resized + padded images, normalised on the device
            if x.dim() != 4 or x.shape[3] != 3 or not x.is_cuda or x.shape[1] % 32 or x.shape[2] % 32:
                raise FdError("uint8 input must be a CUDA [B, H, W, 3] tensor with H, W multiples of 32")
            B, H, W, _ = x.shape
            return self._get_plan(("model_u8", B, H, W, str(x.device)), lambda: self.build_plan(B, H, W, x.device, "u8"))
        self._check_image(x)
        B, _, H, W = x.shape
        return self._get_plan(("model", B, H, W, str(x.device)), lambda: self.build_plan(B, H, W, x.device))

    def detect(self, x: torch.Tensor, head, clip: bool = True):
        """model(x) -> FCOSHead -> ClipBoxes as ONE plan (engine.add_postprocess): the padded detections of FCOSHead.detect_padded --
        (scores [B,K], classes [B,K] int64, boxes [B,K,4] clipped to the image, counts [B] int32) -- with no Python-side work between the
        launches; with `use_graph` the whole detection is one HIP graph launch.  The reference's test loop (test.py:202-223: model,
        FCOSHead(0.05, 0.6, 1000, strides), ClipBoxes at batch 1) is launch-bound; this is its latency path.  Results are plan-owned
        (overwritten by the next call with the same shape)."""
        self._check_eval()
        self._check_image(x)
        key = ("det", float(head.score), float(head.nms_threshold), int(head.max_box), tuple(int(v) for v in head.strides), bool(clip))
        plan = self.plan_for(x, slot=key)
        if getattr(plan, "dets", None) is None:
            engine.add_postprocess(plan, plan.outs, plan.segs, head.strides, head.score, head.nms_threshold, head.max_box, (x.shape[2], x.shape[3]), clip)
        plan.image_ref[0] = x.contiguous()
        if self.use_graph and plan.graph is None:
            plan.capture_graph()
        plan.run()
        return plan.dets

    def forward_images(self, images, events=None):
        """Input pipeline tail on the device: a list of RESIZED uint8 [h_n, w_n, 3] CUDA images (cv2.resize
        stays on the host, dataset/voc.py:117-126) -> pad-to-32, pad to the batch maximum, ToTensor + Normalize in ONE
        launch straight into the stem's input layout -> the model.  Returns what model(batch_imgs) returns; the canvas
        size is `plan.canvas_hw` of `plan_for(images)`."""
        plan = self.plan_for(list(images))
        plan.image_ref[0] = [t.contiguous() for t in images]
        plan.run(events)
        return self.outputs_of(plan)

    def _check_eval(self) -> None:
        if self.training:
            raise FdError("the HIP plan implements the frozen-BN inference forward; call model.eval() first")

    freeze_all_bn = False   # opt-in: keep EVERY BatchNorm2d (not only the backbone's) on its running statistics in train()

    def train(self, mode: bool = True):
        """nn.Module.train, except that the BACKBONE's BatchNorm layers stay in eval mode when the constructor froze them
        (pretrained statistics: the reference freezes them at construction, HISFcos.py:57-68, but its unguarded
        model.train(), train.py:151, silently un-freezes the statistics; SURVEY.md §2 C3 / §8e recommends keeping the
        backbone frozen).  Every other BatchNorm -- the randomly initialised FPN ones (HisBlock bn1-4, fpn.gn1-3) --
        follows nn.Module.train() exactly as in the reference: batch statistics, running-stat updates, SyncBN under DDP,
        affine parameters frozen by the constructor.  `model.freeze_all_bn = True` pins those too (every BN folded into the
        HIP conv epilogues: the fastest training step, for fine-tuning from a converged checkpoint)."""
        if mode != self.training:
            self._plans.clear()      # a training step rewrites weights and running statistics in place (HIP launches do not bump
                                     # tensor version counters): plans are rebuilt on the next eval forward
        super().train(mode)
        if mode and getattr(self, "backbone_freeze", False):
            root = self if self.freeze_all_bn else getattr(self, "backbone", None)
            for m in (root.modules() if root is not None else ()):
                if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
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
