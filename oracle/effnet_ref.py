"""oracle/effnet_ref.py — CPU restatement (torch fp32, functional) of the EfficientNet trunk the reference wraps.

TEST INFRASTRUCTURE ONLY (imported by tests/, never by pytorch_object_detection_amd/).

**PARITY UNPINNED, THIRD-PARTY.**  The reference's `EfficientNetV1` (model/backbone/efficientnetv1.py:11-26) is a thin
wrapper: `EfficientNet.from_pretrained(name)`, `set_swish(memory_efficient=False)`, `extract_endpoints(x)` and the list
`[reduction_1 .. reduction_5]`.  All arithmetic lives in the pip package `efficientnet_pytorch`, pinned 0.7.1 by the
reference (README.md:16) and absent from /root/reference, from this image and from the network; the reference holds no
fixture for it.  This file restates the published algorithm of efficientnet_pytorch 0.7.1 (model.py / utils.py):

  * compound scaling table `efficientnet_params` (width, depth, resolution, dropout) and the seven base stages
    r1_k3_s11_e1_i32_o16_se0.25 ... r1_k3_s11_e6_i192_o320_se0.25; `round_filters` (divisor 8, +8 when rounding lost
    more than 10 %), `round_repeats` (ceil);
  * `Conv2dStaticSamePadding`: `from_pretrained` builds the net with `image_size = resolution of the model name`
    (300 for b3), so every conv carries a ZeroPad2d computed ONCE for that nominal size — TF "SAME" for the nominal
    size, applied whatever the real input is: pad = max((ceil(i/s)-1)*s + (k-1) + 1 - i, 0), split (pad//2, pad-pad//2);
    the nominal size is propagated ceil(i/s) through the stem and every block;
  * MBConvBlock.forward: [expand 1x1 -> BN -> swish] (expand_ratio != 1) -> depthwise kxk stride s -> BN -> swish ->
    SE (adaptive_avg_pool2d 1 -> 1x1 reduce + bias -> swish -> 1x1 expand + bias -> sigmoid -> multiply; squeezed width
    max(1, int(block_input_filters * 0.25))) -> project 1x1 -> BN -> (+ inputs when stride 1 and in == out filters;
    drop_connect is the identity in eval);
  * BatchNorm eps 1e-3 (batch_norm_epsilon), swish(x) = x * sigmoid(x);
  * extract_endpoints: stem -> blocks; `reduction_i` is the activation in front of each resolution drop, the output of
    the last block (`elif idx == len(blocks) - 1`) and finally the head conv (`reduction_6`, unused by the reference).

State-dict keys are efficientnet_pytorch's (`_conv_stem.weight`, `_bn0.*`, `_blocks.{i}._expand_conv.weight`, `_bn0`,
`_depthwise_conv`, `_bn1`, `_se_reduce`, `_se_expand`, `_project_conv`, `_bn2`, `_conv_head`, `_bn1`, `_fc`) under the
reference's attribute path `backbone.model.` (efficientnetv1.py:21).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-3

# (width, depth, resolution, dropout) — efficientnet_pytorch.utils.efficientnet_params
PARAMS = {
    "efficientnet-b0": (1.0, 1.0, 224, 0.2), "efficientnet-b1": (1.0, 1.1, 240, 0.2),
    "efficientnet-b2": (1.1, 1.2, 260, 0.3), "efficientnet-b3": (1.2, 1.4, 300, 0.3),
    "efficientnet-b4": (1.4, 1.8, 380, 0.4), "efficientnet-b5": (1.6, 2.2, 456, 0.4),
    "efficientnet-b6": (1.8, 2.6, 528, 0.5), "efficientnet-b7": (2.0, 3.1, 600, 0.5),
    "efficientnet-b8": (2.2, 3.6, 672, 0.5), "efficientnet-l2": (4.3, 5.3, 800, 0.5),
}
VALID_MODELS = ("efficientnet-b0", "efficientnet-b1", "efficientnet-b2", "efficientnet-b3", "efficientnet-b4",
                "efficientnet-b5", "efficientnet-b6", "efficientnet-b7", "efficientnet-b8", "efficientnet-l2")
# (repeats, kernel, stride, expand, in, out) — se_ratio 0.25, id_skip True everywhere
BASE_STAGES = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
               (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))


def round_filters(filters: int, width: float, divisor: int = 8) -> int:
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats: int, depth: float) -> int:
    return int(math.ceil(depth * repeats))


def block_table(name: str):
    """-> (stem_out, [(kernel, stride, expand, in, out, nominal_input_size)], head_out, nominal_sizes)."""
    width, depth, res, _ = PARAMS[name]
    size = int(math.ceil(res / 2))          # after the stride-2 stem
    blocks = []
    for rep, k, s, e, i, o in BASE_STAGES:
        i, o, rep = round_filters(i, width), round_filters(o, width), round_repeats(rep, depth)
        for r in range(rep):
            blocks.append((k, s if r == 0 else 1, e, i if r == 0 else o, o, size))
            if r == 0:
                size = int(math.ceil(size / s))
    return round_filters(32, width), blocks, round_filters(1280, width), res


def same_pad(nominal: int, k: int, s: int) -> Tuple[int, int]:
    """Conv2dStaticSamePadding for a square nominal image: (pad_before, pad_after) along one axis."""
    out = int(math.ceil(nominal / s))
    pad = max((out - 1) * s + (k - 1) + 1 - nominal, 0)
    return pad // 2, pad - pad // 2


def _bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False,
                        0.0, BN_EPS)


def _swish(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def _conv_same(sd: SD, p: str, x: torch.Tensor, nominal: int, stride: int = 1, groups: int = 1) -> torch.Tensor:
    w = sd[p + ".weight"]
    lo, hi = same_pad(nominal, w.shape[-1], stride)
    if lo or hi:
        x = F.pad(x, (lo, hi, lo, hi))
    return F.conv2d(x, w, sd.get(p + ".bias"), stride, 0, 1, groups)


def mbconv(sd: SD, p: str, x: torch.Tensor, k: int, s: int, e: int, cin: int, cout: int, nominal: int) -> torch.Tensor:
    inputs = x
    if e != 1:
        x = _swish(_bn(sd, p + "._bn0", _conv_same(sd, p + "._expand_conv", x, nominal)))
    x = _swish(_bn(sd, p + "._bn1", _conv_same(sd, p + "._depthwise_conv", x, nominal, s, groups=x.shape[1])))
    sq = F.adaptive_avg_pool2d(x, 1)
    sq = _swish(F.conv2d(sq, sd[p + "._se_reduce.weight"], sd[p + "._se_reduce.bias"]))
    sq = F.conv2d(sq, sd[p + "._se_expand.weight"], sd[p + "._se_expand.bias"])
    x = torch.sigmoid(sq) * x
    x = _bn(sd, p + "._bn2", _conv_same(sd, p + "._project_conv", x, int(math.ceil(nominal / s))))
    if s == 1 and cin == cout:
        x = x + inputs
    return x


def extract_endpoints(sd: SD, x: torch.Tensor, name: str, prefix: str = "backbone.model.") -> Dict[str, torch.Tensor]:
    """EfficientNet.extract_endpoints (efficientnet_pytorch 0.7.1) in eval mode."""
    stem_out, blocks, head_out, res = block_table(name)
    end: Dict[str, torch.Tensor] = {}
    x = _swish(_bn(sd, prefix + "_bn0", _conv_same(sd, prefix + "_conv_stem", x, res, 2)))
    prev = x
    for idx, (k, s, e, cin, cout, nominal) in enumerate(blocks):
        x = mbconv(sd, f"{prefix}_blocks.{idx}", x, k, s, e, cin, cout, nominal)
        if prev.size(2) > x.size(2):
            end[f"reduction_{len(end) + 1}"] = prev
        elif idx == len(blocks) - 1:
            end[f"reduction_{len(end) + 1}"] = x
        prev = x
    x = _swish(_bn(sd, prefix + "_bn1", F.conv2d(x, sd[prefix + "_conv_head.weight"])))
    end[f"reduction_{len(end) + 1}"] = x
    return end


def efficientnet_endpoints5(sd: SD, x: torch.Tensor, backbone_number: int, prefix: str = "backbone.model.") -> List[torch.Tensor]:
    """EfficientNetV1.forward, efficientnetv1.py:24-26: [reduction_1, ..., reduction_5]."""
    e = extract_endpoints(sd, x, VALID_MODELS[backbone_number], prefix)
    return [e[f"reduction_{i}"] for i in range(1, 6)]


def fcos_effnet_forward(sd: SD, x: torch.Tensor, backbone_number: int):
    """FCOS(efficientnet=True) as the authors ran it (Result/propose_giou_50_61.1:77-99 "ef-B0"): the FPN takes the three
    deepest endpoints (reduction_3/4/5 = strides 8/16/32) as (C3, C4, C5).  As shipped Fcos.py:78 unpacks the five-entry
    list into three names and raises; this is the repaired call the product implements."""
    from . import torch_ref as R
    f = efficientnet_endpoints5(sd, x, backbone_number)
    return R.fcos_head(sd, R.fcos_fpn(sd, f[2:]))
