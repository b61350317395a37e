"""HISFCOS on MI355X — same constructor signatures, attribute names and state_dict keys as the reference's
model/od/HISFcos.py (HalfInvertedStageFCOS :45-74, HisBlock :77-112, HalfInvertedStageFPN :115-179,
HISFCOSHead :182-229); forward() runs a compiled plan of hand-written HIP kernels (engine.py)."""
from __future__ import annotations

from typing import List

import numpy as np
import torch
import torch.nn as nn

from ... import engine
from ..._lib import FdError, Segs
from ...ops import Rows
from ..backbone.resnet50 import ResNet50v2
from ..modules.modules import DepthWiseConv2d, PointWiseConv, ScaleExp, SEBlock
from ._planned import PlannedModule, copy_in_nchw, pyramid_out


class HisBlock(nn.Module):
    def __init__(self, feature: int, beta: int = 4, d_rate: int = 2):
        super().__init__()
        half = feature // 2
        self.conv1 = nn.Conv2d(feature, half, 1, 1, 'same')
        self.conv2 = nn.Conv2d(feature, half, 1, 1, 'same')
        self.conv3 = nn.Conv2d(feature, half, 3, 1, 'same', bias=False)
        self.conv4 = nn.Conv2d(feature, feature, 3, 1, 'same', d_rate, bias=False)
        self.conv1_1 = DepthWiseConv2d(half, 3, 1, False)
        self.conv1_2 = SEBlock(half, beta)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm2d(half), nn.BatchNorm2d(half), nn.BatchNorm2d(half)
        self.bn4 = nn.BatchNorm2d(feature)


class HalfInvertedStageFPN(PlannedModule):
    def __init__(self, feature_map: List[int], feature: int):
        super().__init__()
        self.tf1 = nn.Conv2d(feature_map[2], feature, 1, 1, 'same', bias=False)
        self.tf2 = nn.Conv2d(feature_map[1], feature, 1, 1, 'same', bias=False)
        self.tf3 = nn.Conv2d(feature_map[0], feature, 1, 1, 'same', bias=False)
        for i in range(1, 8):
            setattr(self, f"HisBlock{i}", HisBlock(feature, 4, 2))
        # named gn* but BatchNorm2d, as in the reference (HISFcos.py:137-142); gn3 exists and is never used
        self.gn1, self.gn2, self.gn3 = nn.BatchNorm2d(feature), nn.BatchNorm2d(feature), nn.BatchNorm2d(feature)

    def forward(self, x):
        """(C3, C4, C5) NCHW CUDA tensors -> PyramidOut of 5 NCHW-shaped maps (strides 8..128)."""
        self._check_eval()
        c3, c4, c5 = x
        key = ("fpn",) + tuple(tuple(t.shape) for t in x) + (str(c3.device),)

        def build():
            plan = engine.Plan(c3.device, self.conv_precision)
            ins = []
            for t in (c3, c4, c5):
                B, C, H, W = t.shape
                ins.append((plan.pool.get(B * H * W, C), Segs.make(B, [(H, W)])))
            pyr, segs = engine.build_his_fpn(plan, self, ins)
            return plan, ins, pyr, segs

        plan, ins, pyr, segs = self._get_plan(key, build)
        for (r, s), t in zip(ins, (c3, c4, c5)):
            copy_in_nchw(r, s, 0, t)
        plan.run()
        return pyramid_out(pyr, segs)


class HISFCOSHead(PlannedModule):
    def __init__(self, feature: int, num_class: int, prior: float = 0.01):
        super().__init__()
        self.class_num, self.prior = num_class, prior
        self.pw1 = PointWiseConv(feature, 2 * feature)
        self.pw2 = PointWiseConv(2 * feature, feature, bs=True)
        self.dw1 = DepthWiseConv2d(2 * feature, 3)
        self.gn1, self.gn2 = nn.GroupNorm(32, 2 * feature), nn.GroupNorm(32, 2 * feature)
        self.cls_conv = nn.Sequential(nn.Conv2d(feature, feature, 3, padding='same', bias=False),
                                      nn.GroupNorm(32, feature), nn.ReLU(True))
        self.reg_conv = nn.Sequential(nn.Conv2d(feature, feature, 3, padding='same', bias=False),
                                      nn.GroupNorm(32, feature), nn.ReLU(True))
        self.cls_logits = nn.Conv2d(feature, num_class, 3, padding=1)
        self.cnt_logits = nn.Conv2d(feature, 1, 3, padding=1)
        self.reg_pred = nn.Conv2d(feature, 4, 3, padding=1)
        nn.init.constant_(self.cls_logits.bias, -np.log((1 - prior) / prior))
        self.scale_exp = nn.ModuleList([ScaleExp(1.2) for _ in range(5)])

    def forward(self, inputs):
        """5 pyramid maps (PyramidOut or a list of NCHW CUDA tensors) -> (cls_logits, cnt_logits, reg_preds)."""
        self._check_eval()
        shapes = tuple(tuple(t.shape) for t in inputs)
        key = ("head",) + shapes + (str(inputs[0].device),)

        def build():
            plan = engine.Plan(inputs[0].device, self.conv_precision)
            B, C = shapes[0][0], shapes[0][1]
            segs = Segs.make(B, [(s[2], s[3]) for s in shapes])
            pyr = plan.pool.get(segs.rows, C)
            outs = engine.build_his_head(plan, self, pyr, segs)
            return plan, pyr, segs, outs

        plan, pyr, segs, outs = self._get_plan(key, build)
        for i, t in enumerate(inputs):
            copy_in_nchw(pyr, segs, i, t)
        plan.run()
        return tuple(pyramid_out(o, segs) for o in outs)


class HalfInvertedStageFCOS(PlannedModule):
    def __init__(self, feature_map: List[int], num_classes: int, feature: int, bn_freeze: bool = True):
        super().__init__()
        self.backbone = ResNet50v2()
        self.backbone_freeze = bn_freeze
        self.fpn = HalfInvertedStageFPN(feature_map, feature)
        self.head = HISFCOSHead(feature, num_classes, 0.01)
        if self.backbone_freeze:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
                    for p in m.parameters():
                        p.requires_grad = False
            self.backbone.freeze_stages(1)

    def build_plan(self, B: int, H: int, W: int, device):
        plan = engine.Plan(device, self.conv_precision)
        plan.image_ref = [None]
        feats = engine.build_resnet50(plan, self.backbone.trunk, B, H, W, plan.image_ref)
        plan.marks["backbone_end"] = (0, len(plan.steps))
        pyr, segs = engine.build_his_fpn(plan, self.fpn, feats)
        for r, _ in feats:
            plan.pool.put(r)
        plan.marks["fpn_end"] = (0, len(plan.steps))
        outs = engine.build_his_head(plan, self.head, pyr, segs)
        plan.outs, plan.segs = outs, segs
        return plan

    def plan_for(self, x: torch.Tensor):
        self._check_image(x)
        self._check_eval()
        B, _, H, W = x.shape
        return self._get_plan(("model", B, H, W, str(x.device)), lambda: self.build_plan(B, H, W, x.device))

    def forward(self, x: torch.Tensor, events=None):
        """[B,3,H,W] fp32 CUDA -> (cls_logits, cnt_logits, reg_preds), each a list of 5 NCHW-shaped tensors
        (strides 8..128; reference HISFcos.py:70-74).  The tensors are views of plan-owned buffers and are
        overwritten by the next forward of the same shape."""
        plan = self.plan_for(x)
        plan.image_ref[0] = x.contiguous()
        plan.run(events)
        return tuple(pyramid_out(o, plan.segs) for o in plan.outs)
