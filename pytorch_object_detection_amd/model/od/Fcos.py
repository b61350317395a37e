"""FCOS-R50 baseline on MI355X — mirrors the reference's model/od/Fcos.py (FCOS :12-58,
FeaturePyramidNetwork :61-91, HeadFCOS :94-133): same constructors and state_dict keys, HIP plan forward."""
from __future__ import annotations

from typing import List

import numpy as np
import torch
import torch.nn as nn

from ... import engine
from ..._lib import FdError, Segs
import torch.nn.functional as F

from ...ops import ACT_RELU
from ... import train_ops as T
from ...train_ops import conv2d as tconv, conv_bn_act as cba
from ..backbone.resnet50 import ResNet50, trunk_train_forward
from ..modules.modules import ScaleExp, init_conv_kaiming, init_conv_random_normal
from ._planned import PlannedModule, copy_in_nchw, pyramid_out


class FeaturePyramidNetwork(PlannedModule):
    def __init__(self, in_channel: List[int], feature: int = 256):
        super().__init__()
        self.P5 = nn.Conv2d(in_channel[0], feature, 1, 1, 'same')
        self.P4 = nn.Conv2d(in_channel[1], feature, 1, 1, 'same')
        self.P3 = nn.Conv2d(in_channel[2], feature, 1, 1, 'same')
        self.P5_c1 = nn.Conv2d(feature, feature, 3, 1, 'same')
        self.P4_c1 = nn.Conv2d(feature, feature, 3, 1, 'same')
        self.P3_c1 = nn.Conv2d(feature, feature, 3, 1, 'same')
        self.P6_c1 = nn.Conv2d(feature, feature, 3, 2, 1)
        self.P7_c1 = nn.Conv2d(feature, feature, 3, 2, 1)
        self.apply(init_conv_kaiming)

    def train_forward(self, x):
        c3, c4, c5 = x
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")  # noqa: E731
        p5 = tconv(self.P5, c5)
        p4 = tconv(self.P4_c1, up(p5) + tconv(self.P4, c4))
        p3 = tconv(self.P3_c1, up(p4) + tconv(self.P3, c3))
        p5 = tconv(self.P5_c1, p5)
        p6 = cba(self.P6_c1, None, p5, ACT_RELU)   # the reference's in-place ReLU rectifies the returned P6 (Fcos.py:90)
        return p3, p4, p5, p6, tconv(self.P7_c1, p6)

    def forward(self, x):
        if self.training:
            return self.train_forward(x)
        c3, c4, c5 = x
        key = ("FPN",) + tuple(tuple(t.shape) for t in x) + (str(c3.device),)

        def build():
            plan = engine.Plan(c3.device, self.conv_precision)
            ins = []
            for t in (c3, c4, c5):
                B, C, H, W = t.shape
                full, view = engine.padded_input(plan, B * H * W, C)
                ins.append((full, Segs.make(B, [(H, W)]), view))
            pyr, segs = engine.build_fcos_fpn(plan, self, [(r, s) for r, s, _ in ins])
            return plan, ins, pyr, segs

        plan, ins, pyr, segs = self._get_plan(key, build)
        for (_, s, view), t in zip(ins, (c3, c4, c5)):
            copy_in_nchw(view, s, 0, t)
        plan.run()
        return pyramid_out(pyr, segs)


class HeadFCOS(PlannedModule):
    def __init__(self, feature: int, num_class: int, prior: float = 0.01):
        super().__init__()
        self.class_num, self.prior = num_class, prior
        cls_branch, reg_branch = [], []
        for _ in range(4):
            cls_branch += [nn.Conv2d(feature, feature, 3, padding=1, bias=False), nn.GroupNorm(32, feature), nn.ReLU(True)]
            reg_branch += [nn.Conv2d(feature, feature, 3, padding=1, bias=False), nn.GroupNorm(32, feature), nn.ReLU(True)]
        self.cls_branch = nn.Sequential(*cls_branch)
        self.reg_branch = nn.Sequential(*reg_branch)
        self.cls_logits = nn.Conv2d(feature, num_class, 3, padding=1)
        self.cnt_logits = nn.Conv2d(feature, 1, 3, padding=1)
        self.reg_pred = nn.Conv2d(feature, 4, 3, padding=1)
        self.apply(init_conv_random_normal)
        nn.init.constant_(self.cls_logits.bias, -np.log((1 - prior) / prior))
        self.scale_exp = nn.ModuleList([ScaleExp(1.0) for _ in range(5)])

    def train_forward(self, inputs):
        """Training-time autograd forward: the five levels as one rows buffer, every tower layer (3x3 conv, GroupNorm +
        ReLU) and the predictors one HIP launch over the whole pyramid, forward and backward (train_ops)."""
        x0 = inputs[0]
        gns = [self.cls_branch[3 * k + 1] for k in range(4)] + [self.reg_branch[3 * k + 1] for k in range(4)]
        if not T.covered(self.cls_branch[0], None, x0) or not all(T._gn_ok(g, x0) for g in gns):
            return self._train_forward_stock(inputs)
        f, segs = T.pyramid_rows(inputs)
        c, r = f, f
        for k in range(4):
            c = T.groupnorm_rows(self.cls_branch[3 * k + 1], T.conv_rows(self.cls_branch[3 * k], c, segs), segs,
                                 self.cls_branch[3 * k + 2])
            r = T.groupnorm_rows(self.reg_branch[3 * k + 1], T.conv_rows(self.reg_branch[3 * k], r, segs), segs,
                                 self.reg_branch[3 * k + 2])
        cls = T.conv_rows(self.cls_logits, c, segs, pad_out=True)
        rc = T.conv_rows(T.MergedConv(self.reg_pred, self.cnt_logits), r, segs, pad_out=True)  # [:, :4] boxes, [:, 4] centre-ness
        cls_l = T.pyramid_split(cls, segs)
        cnt_l = T.pyramid_split(rc[:, 4:5], segs)
        reg_l = [torch.exp(t * self.scale_exp[i].scale) for i, t in enumerate(T.pyramid_split(rc[:, :4], segs))]
        return cls_l, cnt_l, reg_l

    def _train_forward_stock(self, inputs):
        T.stock_fallback("the HeadFCOS GroupNorm layers (widths outside the rows kernels)")
        cls_l, cnt_l, reg_l = [], [], []
        for i, f in enumerate(inputs):
            c, r = f, f
            for k in range(4):
                c = self.cls_branch[3 * k + 2](self.cls_branch[3 * k + 1](tconv(self.cls_branch[3 * k], c)))
                r = self.reg_branch[3 * k + 2](self.reg_branch[3 * k + 1](tconv(self.reg_branch[3 * k], r)))
            cls_l.append(tconv(self.cls_logits, c))
            cnt_l.append(self.cnt_logits(r))
            reg_l.append(torch.exp(tconv(self.reg_pred, r) * self.scale_exp[i].scale))
        return cls_l, cnt_l, reg_l

    def forward(self, inputs):
        if self.training:
            return self.train_forward(inputs)
        shapes = tuple(tuple(t.shape) for t in inputs)
        key = ("head",) + shapes + (str(inputs[0].device),)

        def build():
            plan = engine.Plan(inputs[0].device, self.conv_precision)
            B, C = shapes[0][0], shapes[0][1]
            segs = Segs.make(B, [(s[2], s[3]) for s in shapes])
            pyr = plan.pool.get(segs.rows, C)
            outs = engine.build_fcos_head(plan, self, pyr, segs)
            return plan, pyr, segs, outs

        plan, pyr, segs, outs = self._get_plan(key, build)
        for i, t in enumerate(inputs):
            copy_in_nchw(pyr, segs, i, t)
        plan.run()
        return tuple(pyramid_out(o, segs) for o in outs)


class FCOS(PlannedModule):
    """FCOS(in_channel [C5, C4, C3 widths], num_class, feature, freeze_bn, efficientnet) — reference Fcos.py:12-58.

    `efficientnet=True` selects the EfficientNet trunk.  As shipped the reference hard-codes `EfficientNetV1(0)` (B0,
    Fcos.py:31-32) and then unpacks the wrapper's FIVE endpoints into three names (Fcos.py:78), which raises; the authors'
    recorded runs ("ef-B0", Result/propose_giou_50_61.1:77-99) imply the three deepest endpoints feed the FPN.  Here:
    `backbone_number` (extra keyword, default 0 = the reference's B0; 3 = BASELINE Cfg5's B3) picks the model and
    reduction_3/4/5 (strides 8/16/32) go to the FPN as (C3, C4, C5); in_channel must be their widths reversed
    (B0 [320, 112, 40], B3 [384, 136, 48])."""

    def __init__(self, in_channel: List[int], num_class: int, feature: int, freeze_bn: bool = True,
                 efficientnet: bool = False, backbone_number: int = 0):
        super().__init__()
        self.efficientnet = bool(efficientnet)
        if efficientnet:
            from ..backbone.efficientnetv1 import EfficientNetV1    # (lazy: that module imports this package's _planned)
            self.backbone = EfficientNetV1(backbone_number)
            want = self.backbone.endpoint_channels[:1:-1]
            if list(in_channel) != want:
                raise FdError(f"FCOS(efficientnet=True, backbone_number={backbone_number}) needs in_channel={want} "
                              f"(reduction_5/4/3 widths), got {list(in_channel)}")
            blocks = list(self.backbone.model._blocks)
            # widest stride-2 activation: the stem output or the first stride-2 block's expanded input (plan_batch_limit)
            self._largest_map_channels = max([self.backbone.model._conv_stem.out_channels] +
                                             [b._depthwise_conv.in_channels for b in blocks[:next(i for i, b in enumerate(blocks) if b.stride > 1) + 1]])
        else:
            self.backbone = ResNet50(3)
        self.FPN = FeaturePyramidNetwork(in_channel, feature)
        self.head = HeadFCOS(feature, num_class, 0.01)
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
        if self.efficientnet:
            feats = engine.build_efficientnet(plan, self.backbone.model, B, H, W, plan.image_ref, keep=(2, 3, 4))[2:]
        else:
            feats = engine.build_resnet50(plan, self.backbone.trunk, B, H, W, plan.image_ref)
        pyr, segs = engine.build_fcos_fpn(plan, self.FPN, feats)
        for r, _ in feats:
            plan.pool.put(r)
        outs = engine.build_fcos_head(plan, self.head, pyr, segs)
        plan.outs, plan.segs = outs, segs
        return plan

    def forward(self, x: torch.Tensor, events=None):
        if self.training:
            self._check_train_input(x)
            if self.efficientnet:
                raise FdError("FCOS(efficientnet=True) is inference-only on the HIP path: the MBConv trunk has no backward "
                              "kernels (the reference's train.py:92-97 never builds it); call model.eval()")
            T.PACKS.refresh()        # every parameter's packed conv weights for this step, one launch
            return self.head.train_forward(self.FPN.train_forward(trunk_train_forward(self.backbone.trunk, x)))
        chunk = self.plan_batch_limit(x)
        if x.shape[0] > chunk:
            return self._forward_chunked(x, chunk)
        plan = self.plan_for(x)
        plan.image_ref[0] = x.contiguous()
        if self.use_graph and plan.graph is None and not events:
            plan.capture_graph()
        plan.run(events)
        return self.outputs_of(plan)
