"""oracle/torch_ref.py — CPU restatement (torch fp32, functional) of the reference's FCOS / HISFCOS hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
the checker / baseline; nothing under pytorch_object_detection_amd/ imports it.

Every function takes a plain state_dict (reference parameter names) and cites the
reference file:line it follows.  Pinned against the imported reference by tests/golden/make_golden.py
(fixtures g8_tiny_fpn_head.npz, g9_*.npz, g5/g6/g7); the ResNet-50 trunk is torchvision arithmetic
(third-party, absent from the reference tree, version unpinned: README.md:5-6) and is restated from the
layer dump the reference itself records (Result/proposed:12-184) — parity for it is "unpinned".
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
_EPS = 1e-5


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
BN_TRAIN_PREFIXES: Tuple[str, ...] = ()   # BatchNorm2d layers whose name starts with one of these run in TRAINING mode


def _bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """nn.BatchNorm2d.  Default: eval (frozen, HISFcos.py:57-68 freeze_bn).  Layers named in BN_TRAIN_PREFIXES use batch
    statistics and update their running statistics in `sd` (momentum 0.1), which is what the reference's model.train()
    (train.py:151) does to every BatchNorm the constructor had put in eval mode."""
    training = any(p.startswith(q) for q in BN_TRAIN_PREFIXES)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training, 0.1 if training else 0.0, _EPS)


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, pad: int = 0, dil: int = 1, groups: int = 1) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, pad, dil, groups)


def _gn(sd: SD, p: str, x: torch.Tensor, groups: int = 32) -> torch.Tensor:
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], _EPS)


# ----------------------------------------------------------------------------------------------
# ResNet-50 v1.5 trunk (torchvision resnet50; reference model/backbone/resnet50.py:68-80, Result/proposed:12-184)
# ----------------------------------------------------------------------------------------------
_R50_LAYERS = ((1, 3, 64, 1), (2, 4, 128, 2), (3, 6, 256, 2), (4, 3, 512, 2))


def _bottleneck(sd: SD, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    out = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, stride=stride, pad=1)))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out))
    if (p + ".downsample.0.weight") in sd:
        x = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride=stride))
    return F.relu(out + x)


def resnet50_c345(sd: SD, x: torch.Tensor, prefix: str = "backbone.extract_feature.") -> Tuple[torch.Tensor, ...]:
    """C3/C4/C5 = outputs of layer2.3 / layer3.5 / layer4.2 (relu_2), resnet50.py:71-80."""
    x = F.relu(_bn(sd, prefix + "bn1", _conv(sd, prefix + "conv1", x, stride=2, pad=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, nblk, _, stride in _R50_LAYERS:
        for b in range(nblk):
            x = _bottleneck(sd, f"{prefix}layer{li}.{b}", x, stride if b == 0 else 1)
        feats.append(x)
    return feats[1], feats[2], feats[3]


# ----------------------------------------------------------------------------------------------
# HISFCOS (reference model/od/HISFcos.py)
# ----------------------------------------------------------------------------------------------
def se_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """SEBlock, model/modules/modules.py:107-121."""
    y = x.mean(dim=(2, 3), keepdim=True)
    y = F.silu(_conv(sd, p + ".excitation.0", y))
    y = torch.sigmoid(_conv(sd, p + ".excitation.2", y))
    return x * y


def his_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """HisBlock.forward, HISFcos.py:95-112."""
    c_half = sd[p + ".conv1.weight"].shape[0]
    x1 = F.silu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    x2 = _conv(sd, p + ".conv2", x)
    u = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv1_1", x1, pad=1, groups=c_half)))
    v = se_block(sd, p + ".conv1_2", x1)
    y = F.relu(_bn(sd, p + ".bn3", _conv(sd, p + ".conv3", torch.cat((u, v), 1), pad=1)))
    out = _bn(sd, p + ".bn4", _conv(sd, p + ".conv4", torch.cat((y, x2), 1), pad=2, dil=2))
    return F.silu(out)


def his_fpn(sd: SD, feats: Sequence[torch.Tensor], p: str = "fpn.") -> Tuple[torch.Tensor, ...]:
    """HalfInvertedStageFPN.forward, HISFcos.py:147-179 (gn2 normalises BOTH the C4 and the C3 lateral)."""
    c3, c4, c5 = feats
    a = F.relu(_bn(sd, p + "gn1", _conv(sd, p + "tf1", c5)))
    x4 = F.max_pool2d(a, 2, 2)
    x5 = F.max_pool2d(x4, 2, 2)
    t3 = his_block(sd, p + "HisBlock1", a)
    l4 = F.relu(_bn(sd, p + "gn2", _conv(sd, p + "tf2", c4)))
    t4 = his_block(sd, p + "HisBlock2", F.interpolate(t3, scale_factor=2.0, mode="nearest") + l4)
    l3 = F.relu(_bn(sd, p + "gn2", _conv(sd, p + "tf3", c3)))
    p3 = his_block(sd, p + "HisBlock3", F.interpolate(t4, scale_factor=2.0, mode="nearest") + l3)
    p4 = his_block(sd, p + "HisBlock4", F.max_pool2d(p3, 2, 2) + t4)
    p5 = his_block(sd, p + "HisBlock5", F.max_pool2d(p4, 2, 2) + t3)
    p6 = his_block(sd, p + "HisBlock6", F.max_pool2d(p5, 2, 2) + x4)
    p7 = his_block(sd, p + "HisBlock7", F.max_pool2d(p6, 2, 2) + x5)
    return p3, p4, p5, p6, p7


def his_head(sd: SD, feats: Sequence[torch.Tensor], p: str = "head."):
    """HISFCOSHead.forward, HISFcos.py:211-229 (cnt_logits is fed from the REG tower)."""
    cls_out, cnt_out, reg_out = [], [], []
    c2 = sd[p + "dw1.weight"].shape[0]
    for i, f in enumerate(feats):
        x = F.relu(_gn(sd, p + "gn1", _conv(sd, p + "pw1", f)))
        x = F.silu(_gn(sd, p + "gn2", _conv(sd, p + "dw1", x, pad=1, groups=c2)))
        z = _conv(sd, p + "pw2", x) + f
        c = F.relu(_gn(sd, p + "cls_conv.1", _conv(sd, p + "cls_conv.0", z, pad=1)))
        r = F.relu(_gn(sd, p + "reg_conv.1", _conv(sd, p + "reg_conv.0", z, pad=1)))
        cls_out.append(_conv(sd, p + "cls_logits", c, pad=1))
        cnt_out.append(_conv(sd, p + "cnt_logits", r, pad=1))
        reg_out.append(torch.exp(_conv(sd, p + "reg_pred", r, pad=1) * sd[f"{p}scale_exp.{i}.scale"]))
    return cls_out, cnt_out, reg_out


def hisfcos_forward(sd: SD, x: torch.Tensor):
    """HalfInvertedStageFCOS.forward, HISFcos.py:70-74."""
    return his_head(sd, his_fpn(sd, resnet50_c345(sd, x)))


# ----------------------------------------------------------------------------------------------
# FCOS baseline (reference model/od/Fcos.py)
# ----------------------------------------------------------------------------------------------
def fcos_fpn(sd: SD, feats: Sequence[torch.Tensor], p: str = "FPN."):
    """FeaturePyramidNetwork.forward, Fcos.py:77-91."""
    c3, c4, c5 = feats
    p5 = _conv(sd, p + "P5", c5)
    p4 = F.interpolate(p5, scale_factor=2.0, mode="nearest") + _conv(sd, p + "P4", c4)
    p4 = _conv(sd, p + "P4_c1", p4, pad=1)
    p3 = F.interpolate(p4, scale_factor=2.0, mode="nearest") + _conv(sd, p + "P3", c3)
    p3 = _conv(sd, p + "P3_c1", p3, pad=1)
    p5 = _conv(sd, p + "P5_c1", p5, pad=1)
    # self.act is nn.ReLU(inplace=True) (Fcos.py:73,90): the P6 map the FPN RETURNS is the rectified one
    p6 = F.relu(_conv(sd, p + "P6_c1", p5, stride=2, pad=1))
    p7 = _conv(sd, p + "P7_c1", p6, stride=2, pad=1)
    return p3, p4, p5, p6, p7


def fcos_head(sd: SD, feats: Sequence[torch.Tensor], p: str = "head."):
    """HeadFCOS.forward, Fcos.py:120-133: 4 x (3x3 conv, GN(32), ReLU) per branch."""
    cls_out, cnt_out, reg_out = [], [], []
    for i, f in enumerate(feats):
        c, r = f, f
        for k in range(4):
            c = F.relu(_gn(sd, f"{p}cls_branch.{3 * k + 1}", _conv(sd, f"{p}cls_branch.{3 * k}", c, pad=1)))
            r = F.relu(_gn(sd, f"{p}reg_branch.{3 * k + 1}", _conv(sd, f"{p}reg_branch.{3 * k}", r, pad=1)))
        cls_out.append(_conv(sd, p + "cls_logits", c, pad=1))
        cnt_out.append(_conv(sd, p + "cnt_logits", r, pad=1))
        reg_out.append(torch.exp(_conv(sd, p + "reg_pred", r, pad=1) * sd[f"{p}scale_exp.{i}.scale"]))
    return cls_out, cnt_out, reg_out


def fcos_forward(sd: SD, x: torch.Tensor):
    """FCOS.forward, Fcos.py:54-58 (backbone ResNet50(3): keys backbone.conv1 / bn1 / layer1..4)."""
    return fcos_head(sd, fcos_fpn(sd, resnet50_c345(sd, x, prefix="backbone.")))


# ----------------------------------------------------------------------------------------------
# MNFCOS (reference model/od/MNFcos.py:11-36,222-297; model/modules/modules.py:195-216) — what config/main.yaml:2 selects
# ----------------------------------------------------------------------------------------------
def mn_block(sd: SD, p: str, x: torch.Tensor, k: int, dil: int) -> torch.Tensor:
    """MNBlock.forward, modules.py:209-216: x + PW2(SiLU(PW1(BN(dilated depthwise(x))))).
    As shipped the depthwise conv is padded with `dilation` (modules.py:203), which keeps the size only for k = 3; for the k = 5 / 7
    blocks of the light-weight FPN the residual add raises (pinned as `fpn_raises` in g10_mnfcos_parts.npz).  Restated with the 'same'
    padding dil * (k - 1) / 2 -- identical to the reference for k = 3 (pinned by g10), the repaired behaviour for k = 5 / 7
    (parity unpinned: there is no reference output)."""
    c = x.shape[1]
    y = _conv(sd, p + "DilatedDepthWiseConv", x, 1, dil * (k - 1) // 2, dil, c)
    y = _bn(sd, p + "BN", y)
    y = F.silu(_conv(sd, p + "PW1", y))
    return x + _conv(sd, p + "PW2", y)


MN_FPN_BLOCKS = {"MNB3": (3, 1), "MNB4": (3, 2), "MNB5": (5, 2), "MNB6": (5, 1), "MNB7": (7, 1)}    # MNFcos.py:230-236 (kernel, dilation)


def mn_fpn(sd: SD, feats: Sequence[torch.Tensor], p: str = "FeaturePyramidNetwork."):
    """LieghtWeightFeaturePyramid_old.forward, MNFcos.py:239-256 (1x1 laterals with bias, nearest x2 upsample + add, 2x2 max-pool)."""
    c3, c4, c5 = feats
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")  # noqa: E731
    blk = lambda n, t: mn_block(sd, p + n + ".", t, *MN_FPN_BLOCKS[n])  # noqa: E731
    p5 = blk("MNB5", _conv(sd, p + "C5PW", c5))
    p4 = blk("MNB4", up(p5) + _conv(sd, p + "C4PW", c4))
    p3 = blk("MNB3", up(p4) + _conv(sd, p + "C3PW", c3))
    p6 = blk("MNB6", F.max_pool2d(p5, 2, 2))
    p7 = blk("MNB7", F.max_pool2d(p6, 2, 2))
    return p3, p4, p5, p6, p7


def mn_head(sd: SD, feats: Sequence[torch.Tensor], p: str = "head."):
    """MNHeadFCOS.forward, MNFcos.py:285-297: two MNBlock(f, f, 3, 2, 2), 3x3 + GroupNorm(32) + SiLU towers, 1x1 predictors, ScaleExp."""
    cls_l, cnt_l, reg_l = [], [], []
    for i, f in enumerate(feats):
        f = mn_block(sd, p + "block2.", mn_block(sd, p + "block1.", f, 3, 2), 3, 2)
        c = F.silu(_gn(sd, p + "cls_conv.1", _conv(sd, p + "cls_conv.0", f, 1, 1)))
        r = F.silu(_gn(sd, p + "reg_conv.1", _conv(sd, p + "reg_conv.0", f, 1, 1)))
        cls_l.append(_conv(sd, p + "cls_logits", c))
        cnt_l.append(_conv(sd, p + "cnt_logits", r))
        reg_l.append(torch.exp(_conv(sd, p + "reg_pred", r) * sd[p + f"scale_exp.{i}.scale"]))
    return cls_l, cnt_l, reg_l


def mnfcos_forward(sd: SD, x: torch.Tensor):
    """MNFCOS.forward, MNFcos.py:32-36 (backbone ResNet50v2: keys backbone.extract_feature.*)."""
    return mn_head(sd, mn_fpn(sd, resnet50_c345(sd, x)))


# ----------------------------------------------------------------------------------------------
# post-processing (reference model/modules/head.py, utill/utills.py) — C restatement via ctypes
# ----------------------------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def clib() -> ctypes.CDLL:
    """oracle/_build/libpostproc_ref.so (built by oracle/Makefile / __graft_entry__.build())."""
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "_build", "libpostproc_ref.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _CLIB = ctypes.CDLL(path)
    return _CLIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def coords_fcos(h: int, w: int, stride: int) -> np.ndarray:
    """coords_origin_fcos, utills.py:58-73."""
    out = np.empty((h * w, 2), np.float32)
    clib().ref_coords(h, w, stride, _p(out))
    return out


def flatten_levels(outs: Sequence[torch.Tensor], nlev: int) -> np.ndarray:
    """reshape_cat_out, head.py:8-26: NCHW -> [B, sum HW, C]; zip() drops levels beyond len(strides)."""
    return torch.cat([o.permute(0, 2, 3, 1).reshape(o.shape[0], -1, o.shape[1]) for o in outs[:nlev]], 1) \
        .contiguous().numpy().astype(np.float32)


def decode(cls: np.ndarray, cnt: np.ndarray, reg: np.ndarray, coords: np.ndarray):
    """head.py:52-66 for a batch: returns scores [B,L], classes [B,L] int32, boxes [B,L,4]."""
    B, L, C = cls.shape
    scores = np.empty((B, L), np.float32)
    classes = np.empty((B, L), np.int32)
    boxes = np.empty((B, L, 4), np.float32)
    coords = np.ascontiguousarray(coords, np.float32)
    for b in range(B):
        clib().ref_decode(_p(np.ascontiguousarray(cls[b])), _p(np.ascontiguousarray(cnt[b].reshape(-1))),
                          _p(np.ascontiguousarray(reg[b])), _p(coords), L, C, _p(scores[b]), _p(classes[b]),
                          _p(boxes[b]))
    return scores, classes, boxes


def topk(scores: np.ndarray, k: int) -> np.ndarray:
    B, L = scores.shape
    idx = np.empty((B, k), np.int32)
    for b in range(B):
        clib().ref_topk(_p(np.ascontiguousarray(scores[b])), L, k, _p(idx[b]))
    return idx


def post_process(scores: np.ndarray, classes: np.ndarray, boxes: np.ndarray, score_thr: float, iou_thr: float):
    """FCOSHead.post_process, head.py:84-102, per image on score-descending rows.  Returns list of keep arrays."""
    lib = clib()
    lib.ref_post_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_float, ctypes.c_double, ctypes.c_void_p]
    keeps = []
    for b in range(scores.shape[0]):
        K = scores.shape[1]
        keep = np.empty(max(K, 1), np.int32)
        n = lib.ref_post_process(_p(np.ascontiguousarray(scores[b], np.float32)),
                                 _p(np.ascontiguousarray(classes[b], np.int64)),
                                 _p(np.ascontiguousarray(boxes[b], np.float32)), K, score_thr, iou_thr, _p(keep))
        keeps.append(keep[:n].copy())
    return keeps


def batched_nms(boxes: np.ndarray, scores: np.ndarray, classes: np.ndarray, iou_thr: float) -> np.ndarray:
    lib = clib()
    lib.ref_batched_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_void_p]
    n = len(scores)
    keep = np.empty(max(n, 1), np.int32)
    k = lib.ref_batched_nms(_p(np.ascontiguousarray(boxes, np.float32)), _p(np.ascontiguousarray(scores, np.float32)),
                            _p(np.ascontiguousarray(classes, np.int64)), n, iou_thr, _p(keep))
    return keep[:k].copy()


def nms(boxes: np.ndarray, scores: np.ndarray, iou_thr: float) -> np.ndarray:
    lib = clib()
    lib.ref_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    n = len(scores)
    keep = np.empty(max(n, 1), np.int32)
    k = lib.ref_nms(_p(np.ascontiguousarray(boxes, np.float32)), _p(np.ascontiguousarray(scores, np.float32)), n,
                    iou_thr, _p(keep))
    return keep[:k].copy()


def box_nms_plus1(boxes: np.ndarray, scores: np.ndarray, thr: float = 0.5, mode: str = "union") -> np.ndarray:
    """DataEncoder._box_nms, utills.py:221-255."""
    lib = clib()
    lib.ref_box_nms_plus1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                      ctypes.c_void_p]
    n = len(scores)
    keep = np.empty(max(n, 1), np.int32)
    k = lib.ref_box_nms_plus1(_p(np.ascontiguousarray(boxes, np.float32)), _p(np.ascontiguousarray(scores, np.float32)),
                              n, thr, {"union": 0, "min": 1}[mode], _p(keep))
    return keep[:k].copy()


def box_nms_plus1_torch(boxes: torch.Tensor, scores: torch.Tensor, thr: float = 0.5, mode: str = "union") -> torch.Tensor:
    """DataEncoder._box_nms (utills.py:221-255) as the reference runs it: a Python `while` over torch tensor ops -- areas with the "+1"
    pixel convention, candidates walked in descending score order, the survivors of each round are those with `ovr <= thr`.  This is the
    form whose CPU time SURVEY 8(d) quotes (104 ms at N = 1000); ref_box_nms_plus1 (postproc_ref.c) is the same rule in plain C.  Pinned to
    the reference's own outputs by the g3 `p*` vectors (tests/test_oracle_golden.py)."""
    if mode not in ("union", "min"):
        raise TypeError(f"Unknown nms mode: {mode}.")
    x1, y1, x2, y2 = boxes.unbind(1)
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.sort(0, descending=True)[1]
    kept: List[int] = []
    while order.numel():
        top = int(order[0])
        kept.append(top)
        rest = order[1:]
        if rest.numel() == 0:
            break
        iw = (torch.clamp(x2[rest], max=float(x2[top])) - torch.clamp(x1[rest], min=float(x1[top])) + 1).clamp(min=0)
        ih = (torch.clamp(y2[rest], max=float(y2[top])) - torch.clamp(y1[rest], min=float(y1[top])) + 1).clamp(min=0)
        inter = iw * ih
        denom = (area[top] + area[rest] - inter) if mode == "union" else area[rest].clamp(max=float(area[top]))
        order = rest[(inter / denom) <= thr]
    return torch.tensor(kept, dtype=torch.long)


def pairwise_iou(a: np.ndarray, b: np.ndarray, plus_one: bool) -> np.ndarray:
    out = np.empty((len(a), len(b)), np.float32)
    clib().ref_pairwise_iou(_p(np.ascontiguousarray(a, np.float32)), _p(np.ascontiguousarray(b, np.float32)), len(a),
                            len(b), int(plus_one), _p(out))
    return out


def clip_boxes(boxes: np.ndarray, img_h: int, img_w: int) -> np.ndarray:
    out = np.ascontiguousarray(boxes, np.float32).copy()
    clib().ref_clip_boxes(_p(out), out.size // 4, img_h, img_w)
    return out


def fcos_detect(outs, strides: Sequence[int], score_thr: float, iou_thr: float, max_box: int,
                img_hw: Tuple[int, int] | None = None):
    """FCOSHead.forward (head.py:52-102) [+ ClipBoxes] on CPU tensors; returns per-image (scores, classes, boxes)."""
    cls_l, cnt_l, reg_l = outs
    nlev = len(strides)
    cls = flatten_levels(cls_l, nlev)
    cnt = flatten_levels(cnt_l, nlev)
    reg = flatten_levels(reg_l, nlev)
    coords = np.concatenate([coords_fcos(o.shape[2], o.shape[3], s) for o, s in zip(cls_l, strides)], 0)
    scores, classes, boxes = decode(cls, cnt, reg, coords)
    k = min(max_box, scores.shape[1])
    idx = topk(scores, k)
    res = []
    for b in range(scores.shape[0]):
        s, c, bx = scores[b][idx[b]], classes[b][idx[b]].astype(np.int64), boxes[b][idx[b]]
        keep = post_process(s[None], c[None], bx[None], score_thr, iou_thr)[0]
        ob = bx[keep]
        if img_hw is not None:
            ob = clip_boxes(ob, *img_hw)
        res.append((s[keep], c[keep], ob))
    return res


# ----------------------------------------------------------------------------------------------
# loss + target assignment (reference model/loss.py, model/modules/head.py:211-316) — torch, autograd-able
# ----------------------------------------------------------------------------------------------
def _flat(preds: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([p.permute(0, 2, 3, 1).reshape(p.shape[0], -1, p.shape[1]) for p in preds], 1)


def iou_loss(pred: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """loss.py:142-152."""
    wh = (torch.min(pred[:, 2:], tgt[:, 2:]) + torch.min(pred[:, :2], tgt[:, :2])).clamp(min=0)
    ov = wh[:, 0] * wh[:, 1]
    a1 = (pred[:, 2] + pred[:, 0]) * (pred[:, 3] + pred[:, 1])
    a2 = (tgt[:, 2] + tgt[:, 0]) * (tgt[:, 3] + tgt[:, 1])
    return (-(ov / (a1 + a2 - ov)).clamp(min=1e-6).log()).sum()


def giou_loss(pred: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """loss.py:155-177."""
    wh = (torch.min(pred[:, 2:], tgt[:, 2:]) + torch.min(pred[:, :2], tgt[:, :2])).clamp(min=0)
    ov = wh[:, 0] * wh[:, 1]
    a1 = (pred[:, 2] + pred[:, 0]) * (pred[:, 3] + pred[:, 1])
    a2 = (tgt[:, 2] + tgt[:, 0]) * (tgt[:, 3] + tgt[:, 1])
    union = a1 + a2 - ov
    whg = (torch.max(pred[:, 2:], tgt[:, 2:]) + torch.max(pred[:, :2], tgt[:, :2])).clamp(min=0)
    g = whg[:, 0] * whg[:, 1]
    giou = ov / union - (g - union) / g.clamp(1e-10)
    return (1.0 - giou).sum()


def focal_loss(logits: torch.Tensor, onehot: torch.Tensor, gamma: float = 2.0, alpha: float = 0.25) -> torch.Tensor:
    """loss.py:180-193 (the 0.99999999995 upper clip is 1.0 in fp32)."""
    p = logits.sigmoid().clip(min=0.000005, max=0.99999999995)
    pt = p * onehot + (1.0 - p) * (1.0 - onehot)
    w = alpha * onehot + (1.0 - alpha) * (1.0 - onehot)
    return (-w * torch.pow(1.0 - pt, gamma) * pt.log()).sum()


def fcos_loss(outs, targets, mode: str = "giou"):
    """FCOSLoss.forward, loss.py:201-215 -> (cls, cnt, reg, total)."""
    cls_l, cnt_l, reg_l = outs
    cls_t, cnt_t, reg_t = targets
    pos = (cnt_t > -1).squeeze(-1)
    num_pos = pos.sum(1).clamp(min=1).float()
    cls, cnt, reg = _flat(cls_l), _flat(cnt_l), _flat(reg_l)
    ncls = cls.shape[-1]
    lc, ln, lr = [], [], []
    for b in range(cls.shape[0]):
        onehot = (torch.arange(1, ncls + 1)[None, :] == cls_t[b]).float()
        lc.append(focal_loss(cls[b], onehot))
        ln.append(F.binary_cross_entropy_with_logits(cnt[b][pos[b]].reshape(-1), cnt_t[b][pos[b]].reshape(-1),
                                                     reduction="sum"))
        fn = giou_loss if mode == "giou" else iou_loss
        lr.append(fn(reg[b][pos[b]], reg_t[b][pos[b]]))
    lc = (torch.stack(lc) / num_pos).mean()
    ln = (torch.stack(ln) / num_pos).mean()
    lr = (torch.stack(lr) / num_pos).mean()
    return lc, ln, lr, lc + ln + lr


def gen_targets(level_hw: Sequence[Tuple[int, int]], strides: Sequence[int], ranges: Sequence[Sequence[int]],
                gt_boxes: torch.Tensor, labels: torch.Tensor, radius: float = 1.5):
    """FCOSGenTargets.forward / generate_target, head.py:218-316."""
    cls_t, cnt_t, reg_t = [], [], []
    B, M = labels.shape
    for (h, w), s, rg in zip(level_hw, strides, ranges):
        xy = torch.from_numpy(coords_fcos(h, w, s))
        x, y = xy[:, 0], xy[:, 1]
        off = torch.stack([x[None, :, None] - gt_boxes[..., 0][:, None, :], y[None, :, None] - gt_boxes[..., 1][:, None, :],
                           gt_boxes[..., 2][:, None, :] - x[None, :, None], gt_boxes[..., 3][:, None, :] - y[None, :, None]], -1)
        area = (off[..., 0] + off[..., 2]) * (off[..., 1] + off[..., 3])
        omin, omax = off.min(-1)[0], off.max(-1)[0]
        cx = (gt_boxes[..., 0] + gt_boxes[..., 2]) / 2
        cy = (gt_boxes[..., 1] + gt_boxes[..., 3]) / 2
        coff = torch.stack([x[None, :, None] - cx[:, None, :], y[None, :, None] - cy[:, None, :],
                            cx[:, None, :] - x[None, :, None], cy[:, None, :] - y[None, :, None]], -1)
        pos = (omin > 0) & (omax > rg[0]) & (omax <= rg[1]) & (coff.max(-1)[0] < s * radius)
        area = area.clone()
        area[~pos] = 99999999
        amin = area.min(-1)[1]
        sel = torch.zeros_like(area, dtype=torch.bool).scatter_(-1, amin.unsqueeze(-1), 1)
        reg = off[sel].reshape(B, -1, 4)
        cls = torch.broadcast_tensors(labels[:, None, :], area.long())[0][sel].reshape(B, -1, 1)
        lr_min, lr_max = torch.min(reg[..., 0], reg[..., 2]), torch.max(reg[..., 0], reg[..., 2])
        tb_min, tb_max = torch.min(reg[..., 1], reg[..., 3]), torch.max(reg[..., 1], reg[..., 3])
        cnt = ((lr_min * tb_min) / (lr_max * tb_max + 1e-10)).sqrt().unsqueeze(-1)
        any_pos = pos.long().sum(-1) >= 1
        cls = cls.clone(); cnt = cnt.clone(); reg = reg.clone()
        cls[~any_pos] = 0
        cnt[~any_pos] = -1
        reg[~any_pos] = -1
        cls_t.append(cls); cnt_t.append(cnt); reg_t.append(reg)
    return torch.cat(cls_t, 1), torch.cat(cnt_t, 1), torch.cat(reg_t, 1)


def normalize_u8(img_u8: np.ndarray, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> np.ndarray:
    """ToTensor + Normalize of dataset/voc.py:57-58,104,155 on an [..., 3] uint8 array -> float32 [..., 3]."""
    a = np.ascontiguousarray(img_u8, np.uint8)
    out = np.empty(a.shape, np.float32)
    clib().ref_normalize_u8(_p(a), a.size // 3, _p(np.asarray(mean, np.float32)), _p(np.asarray(std, np.float32)), _p(out))
    return out


def boxes_rescale_xywh(boxes: np.ndarray, scale: float) -> np.ndarray:
    """Test_coco.py:147-151."""
    lib = clib()
    lib.ref_boxes_rescale_xywh.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float]
    out = np.ascontiguousarray(boxes, np.float32).copy()
    lib.ref_boxes_rescale_xywh(_p(out), out.size // 4, scale)
    return out
