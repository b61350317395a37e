"""MNFCOS on MI355X -- mirrors the reference's model/od/MNFcos.py (MNFCOS :11-36, LieghtWeightFeaturePyramid_old :222-256,
MNHeadFCOS :259-297; the detector config/main.yaml:2 selects and Test_coco.py:201 builds): same constructors and state_dict keys,
HIP plan forward.  SURVEY section 8f n4.

As shipped the reference's forward raises: MNBlock pads its dilated depthwise conv with `dilation` (modules.py:203), which keeps the
map size only for k = 3, and LieghtWeightFeaturePyramid_old uses k = 5 and 7 (MNFcos.py:233-235), so the first residual add
(modules.py:215) fails (tests/golden/g10_mnfcos_parts.npz records it).  Here the padding is 'same' -- identical for the k = 3 blocks
(the whole head, pinned against the reference by g10), the evident intent for the others.  Inference only: the HIP path has no backward
for the dilated depthwise blocks; `train()` + forward raises."""
from __future__ import annotations

from typing import List

import numpy as np
import torch
import torch.nn as nn

from ... import engine
from ..._lib import FdError, Segs
from ..backbone.resnet50 import ResNet50v2
from ..modules.modules import MNBlock, ScaleExp
from ._planned import PlannedModule, copy_in_nchw, pyramid_out


class LieghtWeightFeaturePyramid_old(PlannedModule):
    def __init__(self, in_channel: List[int], feature: int = 128):
        super().__init__()
        self.C5PW = nn.Conv2d(in_channel[0], feature, 1, 1, 'same')
        self.C4PW = nn.Conv2d(in_channel[1], feature, 1, 1, 'same')
        self.C3PW = nn.Conv2d(in_channel[2], feature, 1, 1, 'same')
        self.MNB1_P3 = MNBlock(feature, feature, 3, 2, 2)          # (constructed, never called: MNFcos.py:229)
        self.DownSample_1 = nn.MaxPool2d(2, 2)
        self.DownSample_2 = nn.MaxPool2d(2, 2)
        self.UpSample_1 = nn.Upsample(scale_factor=2)
        self.UpSample_2 = nn.Upsample(scale_factor=2)
        self.MNB7 = MNBlock(feature, feature, 7, 1, 2)
        self.MNB6 = MNBlock(feature, feature, 5, 1, 2)
        self.MNB5 = MNBlock(feature, feature, 5, 2, 2)
        self.MNB4 = MNBlock(feature, feature, 3, 2, 2)
        self.MNB3 = MNBlock(feature, feature, 3, 1, 2)

    def forward(self, x):
        self._check_eval()
        c3, c4, c5 = x
        key = ("FPN",) + tuple(tuple(t.shape) for t in x) + (str(c3.device),)

        def build():
            plan = engine.Plan(c3.device, self.conv_precision)
            ins = []
            for t in (c3, c4, c5):
                B, C, H, W = t.shape
                full, view = engine.padded_input(plan, B * H * W, C)
                ins.append((full, Segs.make(B, [(H, W)]), view))
            pyr, segs = engine.build_mn_fpn(plan, self, [(r, s) for r, s, _ in ins])
            return plan, ins, pyr, segs

        plan, ins, pyr, segs = self._get_plan(key, build)
        for (_, s, view), t in zip(ins, (c3, c4, c5)):
            copy_in_nchw(view, s, 0, t)
        plan.run()
        return pyramid_out(pyr, segs)


class MNHeadFCOS(PlannedModule):
    def __init__(self, feature: int, num_class: int, prior: float = 0.01):
        super().__init__()
        self.class_num, self.prior = num_class, prior
        self.block1 = MNBlock(feature, feature, 3, 2, 2)
        self.block2 = MNBlock(feature, feature, 3, 2, 2)
        self.cls_conv = nn.Sequential(nn.Conv2d(feature, feature, kernel_size=3, padding=1, bias=False), nn.GroupNorm(32, feature), nn.SiLU(True))
        self.reg_conv = nn.Sequential(nn.Conv2d(feature, feature, kernel_size=3, padding=1, bias=False), nn.GroupNorm(32, feature), nn.SiLU(True))
        self.cls_logits = nn.Conv2d(feature, num_class, kernel_size=1)
        self.cnt_logits = nn.Conv2d(feature, 1, kernel_size=1)
        self.reg_pred = nn.Conv2d(feature, 4, kernel_size=1)
        nn.init.constant_(self.cls_logits.bias, -np.log((1 - prior) / prior))
        self.scale_exp = nn.ModuleList([ScaleExp(1.0) for _ in range(5)])

    def forward(self, inputs):
        self._check_eval()
        shapes = tuple(tuple(t.shape) for t in inputs)
        key = ("head",) + shapes + (str(inputs[0].device),)

        def build():
            plan = engine.Plan(inputs[0].device, self.conv_precision)
            B, C = shapes[0][0], shapes[0][1]
            segs = Segs.make(B, [(s[2], s[3]) for s in shapes])
            pyr = plan.pool.get(segs.rows, C)
            outs = engine.build_mn_head(plan, self, pyr, segs)
            return plan, pyr, segs, outs

        plan, pyr, segs, outs = self._get_plan(key, build)
        for i, t in enumerate(inputs):
            copy_in_nchw(pyr, segs, i, t)
        plan.run()
        return tuple(pyramid_out(o, segs) for o in outs)


class MNFCOS(PlannedModule):
    """MNFCOS(in_channel [C5, C4, C3 widths], num_class, feature, freeze_bn) -- reference MNFcos.py:11-36."""

    def __init__(self, in_channel: List[int], num_class: int, feature: int, freeze_bn: bool = True):
        super().__init__()
        self.backbone = ResNet50v2()
        self.FeaturePyramidNetwork = LieghtWeightFeaturePyramid_old(in_channel, feature)
        self.head = MNHeadFCOS(feature, num_class, 0.01)
        self.backbone_freeze = freeze_bn
        if self.backbone_freeze:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
                    for p in m.parameters():
                        p.requires_grad = False

    def build_plan(self, B: int, H: int, W: int, device, input_mode=None):
        plan = engine.Plan(device, self.conv_precision, pair_tuned=getattr(self, "_plan_pair_tuned", False))
        plan.image_ref = [None]
        plan.input_mode, plan.canvas_hw = input_mode, (H, W)
        plan.input_u8 = (self.pixel_mean, self.pixel_std) if input_mode else None
        feats = engine.build_resnet50(plan, self.backbone.trunk, B, H, W, plan.image_ref)
        pyr, segs = engine.build_mn_fpn(plan, self.FeaturePyramidNetwork, feats)
        for r, _ in feats:
            plan.pool.put(r)
        outs = engine.build_mn_head(plan, self.head, pyr, segs)
        plan.outs, plan.segs = outs, segs
        return plan

    def forward(self, x: torch.Tensor, events=None):
        if self.training:
            raise FdError("MNFCOS is inference-only on the HIP path (no backward for the dilated depthwise MNBlocks; as shipped the "
                          "reference's own forward raises, MNFcos.py:233-235 / modules.py:203,215); call model.eval()")
        chunk = self.plan_batch_limit(x)
        if x.shape[0] > chunk:
            return self._forward_chunked(x, chunk)
        plan = self.plan_for(x)
        plan.image_ref[0] = x.contiguous()
        if self.use_graph and plan.graph is None and not events:
            plan.capture_graph()
        plan.run(events)
        return self.outputs_of(plan)
