"""EfficientNet-B0..B7 trunk on MI355X behind the reference's wrapper API (model/backbone/efficientnetv1.py:11-26):
`EfficientNetV1(backbone_number)(x) -> [reduction_1, ..., reduction_5]`.

The reference delegates to the pip package efficientnet_pytorch (pinned 0.7.1, README.md:16: `EfficientNet.from_pretrained`,
`set_swish`, `extract_endpoints`), which is third-party, absent from the reference tree and from this image.  The modules
below only HOLD the parameters under efficientnet_pytorch's names (`model._conv_stem`, `model._bn0`,
`model._blocks.{i}._expand_conv / _bn0 / _depthwise_conv / _bn1 / _se_reduce / _se_expand / _project_conv / _bn2`,
`model._conv_head`, `model._bn1`, `model._fc`), so a reference checkpoint loads with strict=True; the arithmetic
(MBConv: expand 1x1 -> depthwise k x k stride s with static TF-"SAME" padding -> SE -> project 1x1 (+ skip), BN eps 1e-3,
swish) runs in engine.build_efficientnet on the HIP kernels.  There is no network here, hence no pretrained download:
parameters keep torch's default initialisation until a state_dict is loaded (`from_pretrained` in the reference).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn as nn

from ... import engine
from ..._lib import FdError
from ..od._planned import PlannedModule

VALID_MODELS = ('efficientnet-b0', 'efficientnet-b1', 'efficientnet-b2', 'efficientnet-b3', 'efficientnet-b4',
                'efficientnet-b5', 'efficientnet-b6', 'efficientnet-b7', 'efficientnet-b8', 'efficientnet-l2')

# compound-scaling coefficients: name -> (width, depth, nominal resolution, dropout)
_COEFF = {'efficientnet-b0': (1.0, 1.0, 224, 0.2), 'efficientnet-b1': (1.0, 1.1, 240, 0.2),
          'efficientnet-b2': (1.1, 1.2, 260, 0.3), 'efficientnet-b3': (1.2, 1.4, 300, 0.3),
          'efficientnet-b4': (1.4, 1.8, 380, 0.4), 'efficientnet-b5': (1.6, 2.2, 456, 0.4),
          'efficientnet-b6': (1.8, 2.6, 528, 0.5), 'efficientnet-b7': (2.0, 3.1, 600, 0.5),
          'efficientnet-b8': (2.2, 3.6, 672, 0.5), 'efficientnet-l2': (4.3, 5.3, 800, 0.5)}
# the seven B0 stages "r{repeat}_k{kernel}_s{stride}{stride}_e{expand}_i{in}_o{out}_se0.25"
_STAGES = ('r1_k3_s11_e1_i32_o16_se0.25', 'r2_k3_s22_e6_i16_o24_se0.25', 'r2_k5_s22_e6_i24_o40_se0.25',
           'r3_k3_s22_e6_i40_o80_se0.25', 'r3_k5_s11_e6_i80_o112_se0.25', 'r4_k5_s22_e6_i112_o192_se0.25',
           'r1_k3_s11_e6_i192_o320_se0.25')
BN_EPS, BN_MOMENTUM = 1e-3, 1.0 - 0.99


def _scaled_width(filters: int, width: float, divisor: int = 8) -> int:
    f = filters * width
    out = max(divisor, int(f + divisor / 2) // divisor * divisor)
    return int(out + divisor) if out < 0.9 * f else int(out)


def _decode(stage: str) -> dict:
    d = {}
    for tok in stage.split('_'):
        key = 'se' if tok.startswith('se') else tok[0]
        d[key] = tok[len(key):]
    return dict(repeat=int(d['r']), kernel=int(d['k']), stride=int(d['s'][0]), expand=int(d['e']), cin=int(d['i']),
                cout=int(d['o']), se=float(d['se']))


def static_same_padding(nominal: int, kernel: int, stride: int) -> Tuple[int, int]:
    """(before, after) zero padding of efficientnet_pytorch's Conv2dStaticSamePadding for a square nominal image size:
    TF 'SAME' computed once for the size the model was BUILT for, then applied to whatever input arrives."""
    out = int(math.ceil(nominal / stride))
    pad = max((out - 1) * stride + kernel - nominal, 0)
    return pad // 2, pad - pad // 2


class _MBConv(nn.Module):
    """Parameter container of one MBConvBlock; `pad` is the depthwise conv's static (before, after) padding."""

    def __init__(self, kernel: int, stride: int, expand: int, cin: int, cout: int, se_ratio: float, nominal: int):
        super().__init__()
        mid = cin * expand
        self.kernel, self.stride, self.expand, self.cin, self.cout = kernel, stride, expand, cin, cout
        self.pad = static_same_padding(nominal, kernel, stride)
        self.skip = stride == 1 and cin == cout
        if expand != 1:
            self._expand_conv = nn.Conv2d(cin, mid, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, momentum=BN_MOMENTUM, eps=BN_EPS)
        self._depthwise_conv = nn.Conv2d(mid, mid, kernel, stride, groups=mid, bias=False)
        self._bn1 = nn.BatchNorm2d(mid, momentum=BN_MOMENTUM, eps=BN_EPS)
        squeezed = max(1, int(cin * se_ratio))
        self._se_reduce = nn.Conv2d(mid, squeezed, 1)
        self._se_expand = nn.Conv2d(squeezed, mid, 1)
        self._project_conv = nn.Conv2d(mid, cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(cout, momentum=BN_MOMENTUM, eps=BN_EPS)


class _EfficientNet(nn.Module):
    """`EfficientNet.from_name(model_name)` as a parameter container (state_dict keys of efficientnet_pytorch 0.7.1)."""

    def __init__(self, model_name: str):
        super().__init__()
        if model_name not in _COEFF:
            raise FdError(f"unknown EfficientNet '{model_name}'")
        width, depth, res, _ = _COEFF[model_name]
        self.model_name, self.image_size = model_name, res
        stem = _scaled_width(32, width)
        self._conv_stem = nn.Conv2d(3, stem, 3, 2, bias=False)
        self.stem_pad = static_same_padding(res, 3, 2)
        self._bn0 = nn.BatchNorm2d(stem, momentum=BN_MOMENTUM, eps=BN_EPS)
        size = int(math.ceil(res / 2))
        blocks = []
        for st in map(_decode, _STAGES):
            cin, cout = _scaled_width(st['cin'], width), _scaled_width(st['cout'], width)
            for r in range(int(math.ceil(depth * st['repeat']))):
                first = r == 0
                blocks.append(_MBConv(st['kernel'], st['stride'] if first else 1, st['expand'], cin if first else cout, cout,
                                      st['se'], size))
                if first:
                    size = int(math.ceil(size / st['stride']))
        self._blocks = nn.ModuleList(blocks)
        head = _scaled_width(1280, width)
        self._conv_head = nn.Conv2d(blocks[-1].cout, head, 1, bias=False)   # reduction_6: built for checkpoint parity, never run
        self._bn1 = nn.BatchNorm2d(head, momentum=BN_MOMENTUM, eps=BN_EPS)
        self._fc = nn.Linear(head, 1000)


class EfficientNetV1(PlannedModule):
    """Reference model/backbone/efficientnetv1.py:11-26.  forward(x [B,3,H,W] fp32 CUDA) -> list of the five endpoints
    reduction_1..reduction_5 (strides 2, 4, 8, 16, 32; B3: 24, 32, 48, 136, 384 channels) as NCHW-shaped channels-last
    views of plan-owned buffers.  Inference only: BatchNorm runs folded (eval), the HIP path has no MBConv backward."""

    def __init__(self, backbone_number: int, memory_efficient: bool = False):
        super().__init__()
        self.model = _EfficientNet(VALID_MODELS[backbone_number])
        self.memory_efficient = memory_efficient      # swish variant of the reference: same forward arithmetic

    @property
    def endpoint_channels(self) -> List[int]:
        blocks = list(self.model._blocks)
        ch = [b.cout for b, nxt in zip(blocks, blocks[1:]) if nxt.stride > 1]
        return ch + [blocks[-1].cout]

    def build_plan(self, B: int, H: int, W: int, device):
        plan = engine.Plan(device, self.conv_precision)
        plan.image_ref = [None]
        plan.input_u8 = None
        plan.endpoints = engine.build_efficientnet(plan, self.model, B, H, W, plan.image_ref, keep=(0, 1, 2, 3, 4))
        return plan

    def forward(self, x: torch.Tensor):
        if self.training:
            raise FdError("EfficientNet backbones are inference-only on the HIP path (call .eval()); the reference never "
                          "trains them in its shipped scripts (train.py:92-97)")
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32:
            raise FdError("expected a float32 image batch [B, 3, H, W]")
        if not x.is_cuda:
            raise FdError("pytorch_object_detection_amd runs on the GPU only; there is no CPU fallback (got a CPU tensor)")
        B, _, H, W = x.shape
        plan = self._get_plan(("effnet", B, H, W, str(x.device)), lambda: self.build_plan(B, H, W, x.device))
        plan.image_ref[0] = x.contiguous()
        plan.run()
        outs = []
        for rows, segs in plan.endpoints:
            h, w = segs.H[0], segs.W[0]
            outs.append(rows.tensor().view(B, h, w, rows.C).permute(0, 3, 1, 2))
        return outs
