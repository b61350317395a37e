"""tests/golden/make_golden.py — generate the golden fixtures in this directory from the REAL reference.

Run ONLY in the build container (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's absent third-party imports (torchvision, torchinfo, cv2, efficientnet_pytorch, pycocotools,
pytorch_grad_cam, tensorboard) are satisfied by an in-process stub finder; only code whose arithmetic lives in
/root/reference is executed.  The reference source is never copied: this script stores inputs and outputs only
(.npz data).  torchvision.ops.batched_nms (third-party, absent) is NOT executed; the NMS fixtures use the
reference's own live greedy NMS `DataEncoder._box_nms` (utill/utills.py:221-255):
  * plus_one fixtures: `_box_nms` directly;
  * no-plus-one fixtures: integer-valued boxes, for which `_box_nms([x1,y1,x2-1,y2-1])` is exactly the
    no-"+1" rule in fp32; the class offsets follow the authors' own commented restatement
    (model/modules/head.py:104-149: offsets = idxs * (max_coordinate + 1)).
Seeds are fixed; every fixture is a few KB to ~150 KB.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FD_REFERENCE", "/root/reference")
STUBS = {"torchvision", "torchinfo", "efficientnet_pytorch", "cv2", "pycocotools", "pytorch_grad_cam", "tensorboard"}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Stub(self.__name__ + "." + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        raise RuntimeError("third-party stub called: " + self.__name__)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUBS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Finder())
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)
torch.manual_seed(0)

from model import loss as ref_loss  # noqa: E402
from model.modules import head as ref_head  # noqa: E402
from model.od import Fcos as ref_fcos  # noqa: E402
from model.od import HISFcos as ref_his  # noqa: E402
from utill import utills as ref_ut  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KB")


def sd_np(module):
    return {k: v.detach().numpy() for k, v in module.state_dict().items()}


def randomize_bn(module, gen):
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
        if isinstance(m, torch.nn.GroupNorm):
            m.weight.data.copy_(torch.rand(m.num_channels, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_channels, generator=gen) * 0.1)


# ------------------------------------------------------------------ G1 coords
def g1():
    out = {}
    for (h, w) in [(512, 512), (640, 640), (832, 1344)]:
        # level sizes follow the network: /8,/16,/32 then max-pool (floor) twice
        hs = [h // 8, h // 16, h // 32]; ws = [w // 8, w // 16, w // 32]
        hs += [hs[2] // 2, hs[2] // 4]; ws += [ws[2] // 2, ws[2] // 4]
        for lh, lw, s in zip(hs, ws, (8, 16, 32, 64, 128)):
            feat = torch.zeros(1, lh, lw, 1)
            out[f"{h}x{w}_s{s}"] = ref_ut.coords_origin_fcos(feat, s).numpy()
            out[f"{h}x{w}_s{s}_hw"] = np.array([lh, lw])
    save("g1_coords", **out)


# ------------------------------------------------------------------ G2 decode + top-k (post_process input)
def g2():
    gen = torch.Generator().manual_seed(2)
    orig = ref_head.FCOSHead.post_process
    ref_head.FCOSHead.post_process = lambda self, preds: preds  # capture the top-k stage (head.py:82)
    try:
        out = {}
        for ncls in (20, 80):
            sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
            B = 2
            cls = [torch.randn(B, ncls, h, w, generator=gen) * 2 - 3 for h, w in sizes]
            cnt = [torch.randn(B, 1, h, w, generator=gen) for h, w in sizes]
            reg = [torch.exp(torch.randn(B, 4, h, w, generator=gen)) * 8 for h, w in sizes]
            for k, strides in (("s5", [8, 16, 32, 64, 128]), ("s4", [8, 16, 32, 64])):
                head = ref_head.FCOSHead(0.05, 0.6, 100, strides)
                s, c, b = head([cls, cnt, reg])
                out[f"c{ncls}_{k}_scores"] = s.numpy(); out[f"c{ncls}_{k}_classes"] = c.numpy()
                out[f"c{ncls}_{k}_boxes"] = b.numpy(); out[f"c{ncls}_{k}_strides"] = np.array(strides)
            for i in range(5):
                out[f"c{ncls}_cls{i}"] = cls[i].numpy(); out[f"c{ncls}_cnt{i}"] = cnt[i].numpy()
                out[f"c{ncls}_reg{i}"] = reg[i].numpy()
        save("g2_decode_topk", **out)
    finally:
        ref_head.FCOSHead.post_process = orig


# ------------------------------------------------------------------ G3 NMS kept indices
def _rand_boxes(rng, n, integer, crowded=False):
    if crowded:
        nc = max(1, n // 5)
        cxy = rng.uniform(20, 480, (nc, 2)); wh = rng.uniform(16, 120, (nc, 2))
        idx = rng.integers(0, nc, n)
        c = cxy[idx] + rng.normal(0, 4, (n, 2)); s = wh[idx] * rng.uniform(0.85, 1.15, (n, 2))
    else:
        c = rng.uniform(0, 512, (n, 2)); s = np.exp(rng.uniform(np.log(8), np.log(256), (n, 2)))
    b = np.concatenate([c - s / 2, c + s / 2], 1)
    if integer:
        b = np.round(b)
        b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 1)
    return b.astype(np.float32)


def _ref_batched_nms_integer(enc, boxes, scores, classes, thr):
    """coordinate trick (head.py:104-149) + the reference's live greedy NMS on integer boxes."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    off = classes.to(boxes) * (boxes.max() + 1)
    b = boxes + off[:, None]
    assert torch.equal(b, b.round()) and b.abs().max() < 2 ** 23
    bm = b.clone(); bm[:, 2:] -= 1
    return enc._box_nms(bm, scores, threshold=thr).reshape(-1)


def _no_exact_hits(boxes, thr):
    """True when no pair's fp32 IoU equals float32(thr): then `ovr <= thr32` and `(double)ovr > thr` agree."""
    b = torch.from_numpy(boxes)
    lt = torch.max(b[:, None, :2], b[:, :2]); rb = torch.min(b[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0); inter = wh[..., 0] * wh[..., 1]
    a = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iou = inter / (a[:, None] + a - inter)
    return not bool((iou == np.float32(thr)).any())


def g3():
    enc = ref_ut.DataEncoder()
    rng = np.random.default_rng(3)
    out = {}
    case = 0
    for n in (1, 2, 64, 300, 1000):
        for crowded in (False, True):
            for ncls in (1, 20, 80):
                for thr in (0.5, 0.6):
                    boxes = _rand_boxes(rng, n, True, crowded)
                    scores = np.sqrt(rng.uniform(0, 1, n) * rng.uniform(0, 1, n)).astype(np.float32)
                    classes = rng.integers(1, ncls + 1, n).astype(np.int64)
                    off = classes.astype(np.float32) * (boxes.max() + np.float32(1))
                    if not _no_exact_hits(boxes + off[:, None], thr):
                        continue
                    keep = _ref_batched_nms_integer(enc, torch.from_numpy(boxes), torch.from_numpy(scores),
                                                    torch.from_numpy(classes), thr).numpy()
                    out[f"b{case}_boxes"] = boxes; out[f"b{case}_scores"] = scores; out[f"b{case}_classes"] = classes
                    out[f"b{case}_thr"] = np.array(thr); out[f"b{case}_keep"] = keep
                    case += 1
    out["n_batched"] = np.array(case)
    # exact-threshold (IoU == 0.5 exactly: not suppressed) and duplicate boxes (IoU == 1: suppressed)
    boxes = np.array([[0, 0, 2, 1], [0, 0, 1, 1], [10, 10, 20, 20], [10, 10, 20, 20], [10, 10, 20, 30]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.5], np.float32); classes = np.ones(5, np.int64)
    out["edge_boxes"] = boxes; out["edge_scores"] = scores; out["edge_classes"] = classes
    out["edge_keep_0.5"] = _ref_batched_nms_integer(enc, torch.from_numpy(boxes), torch.from_numpy(scores),
                                                     torch.from_numpy(classes), 0.5).numpy()
    # "+1" convention, fractional boxes, both modes: the reference function itself
    case = 0
    for n in (1, 2, 64, 400, 1000):
        for crowded in (False, True):
            for mode in ("union", "min"):
                for thr in (0.5, 0.3):
                    boxes = _rand_boxes(rng, n, False, crowded)
                    scores = rng.uniform(0, 1, n).astype(np.float32)
                    keep = enc._box_nms(torch.from_numpy(boxes), torch.from_numpy(scores), threshold=thr,
                                        mode=mode).reshape(-1).numpy()
                    out[f"p{case}_boxes"] = boxes; out[f"p{case}_scores"] = scores; out[f"p{case}_thr"] = np.array(thr)
                    out[f"p{case}_mode"] = np.array(0 if mode == "union" else 1); out[f"p{case}_keep"] = keep
                    case += 1
    out["n_plus1"] = np.array(case)
    save("g3_nms", **out)


# ------------------------------------------------------------------ G3b end-to-end FCOSHead + ClipBoxes (integer boxes)
def g3b():
    enc = ref_ut.DataEncoder()
    gen = torch.Generator().manual_seed(33)

    def bnms(boxes, scores, idxs, thr):
        assert _no_exact_hits((boxes + (idxs.to(boxes) * (boxes.max() + 1))[:, None]).numpy(), thr)
        return _ref_batched_nms_integer(enc, boxes, scores, idxs, thr)

    ref_head.torchvision.ops.batched_nms = bnms
    out = {}
    sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    for ci, (ncls, thr, sthr) in enumerate([(20, 0.5, 0.05), (80, 0.6, 0.05), (20, 0.6, 0.3)]):
        cls = [torch.randn(1, ncls, h, w, generator=gen) * 2 - 2 for h, w in sizes]
        cnt = [torch.randn(1, 1, h, w, generator=gen) for h, w in sizes]
        reg = [torch.round(torch.exp(torch.randn(1, 4, h, w, generator=gen)) * 12) + 1 for h, w in sizes]
        head = ref_head.FCOSHead(sthr, thr, 200, [8, 16, 32, 64, 128])
        s, c, b = head([cls, cnt, reg])
        img = torch.zeros(1, 3, 128, 128)
        bc = ref_head.ClipBoxes()(img, b.clone())
        out[f"e{ci}_scores"] = s.numpy(); out[f"e{ci}_classes"] = c.numpy(); out[f"e{ci}_boxes"] = b.numpy()
        out[f"e{ci}_clipped"] = bc.numpy(); out[f"e{ci}_cfg"] = np.array([sthr, thr, 200])
        for i in range(5):
            out[f"e{ci}_cls{i}"] = cls[i].numpy(); out[f"e{ci}_cnt{i}"] = cnt[i].numpy(); out[f"e{ci}_reg{i}"] = reg[i].numpy()
    save("g3b_head_end2end", **out)


# ------------------------------------------------------------------ G4 pairwise IoU
def g4():
    enc = ref_ut.DataEncoder()
    rng = np.random.default_rng(4)
    a = _rand_boxes(rng, 37, False); b = _rand_boxes(rng, 53, False, True)
    iou_p1 = enc._box_iou(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    save("g4_pairwise_iou", a=a, b=b, iou_plus1=iou_p1)


# ------------------------------------------------------------------ G5 LTRB IoU / GIoU loss + grad
def g5():
    rng = np.random.default_rng(5)
    out = {}
    for P in (0, 1, 257):
        pred = np.exp(rng.normal(2, 1, (P, 4))).astype(np.float32)
        tgt = np.exp(rng.normal(2, 1, (P, 4))).astype(np.float32)
        if P == 257:
            tgt[:8] = pred[:8]                                   # identical boxes
            pred[8:12] = 1e-3; tgt[8:12] = 200.0                  # tiny IoU -> 1e-6 clamp in iou mode
        for mode, fn in (("iou", ref_loss.iou_loss), ("giou", ref_loss.giou_loss)):
            p = torch.from_numpy(pred).requires_grad_(True)
            l = fn(p, torch.from_numpy(tgt))
            if P:
                l.backward()
            out[f"P{P}_{mode}_loss"] = l.detach().numpy()
            out[f"P{P}_{mode}_grad"] = p.grad.numpy() if P else np.zeros((0, 4), np.float32)
        out[f"P{P}_pred"] = pred; out[f"P{P}_tgt"] = tgt
    save("g5_ltrb_loss", **out)


# ------------------------------------------------------------------ G6/G7 targets + FCOSLoss
def g67():
    gen = torch.Generator().manual_seed(67)
    out = {}
    kat = ref_loss.compute_cnt_loss([torch.ones([2, 1, 4, 4])] * 5, torch.ones([2, 80, 1]), torch.ones([2, 80], dtype=torch.bool))
    out["kat_cnt_loss"] = kat.numpy()          # loss.py:218-221 -> tensor([0.3133, 0.3133])
    sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    strides = [8, 16, 32, 64, 128]
    B, ncls = 2, 20
    gt = torch.tensor([[[10., 12., 60., 70.], [30., 30., 120., 110.], [-1, -1, -1, -1]],
                       [[5., 5., 25., 30.], [0., 0., 127., 127.], [64., 20., 100., 90.]]])
    labels = torch.tensor([[3, 7, -1], [1, 20, 12]])
    for name, ranges in (("voc_his", [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]]),
                         ("voc_fcos", [[-1, 64], [64, 128], [128, 256], [256, 512], [512, 9999999]])):
        cls = [torch.randn(B, ncls, h, w, generator=gen) - 2 for h, w in sizes]
        cnt = [torch.randn(B, 1, h, w, generator=gen) for h, w in sizes]
        reg = [torch.exp(torch.randn(B, 4, h, w, generator=gen)) * 8 for h, w in sizes]
        tg = ref_head.FCOSGenTargets(strides, ranges)([[cls, cnt, reg], gt, labels])
        out[f"{name}_cls_t"] = tg[0].numpy(); out[f"{name}_cnt_t"] = tg[1].numpy(); out[f"{name}_reg_t"] = tg[2].numpy()
        out[f"{name}_ranges"] = np.array(ranges)
        for mode in ("giou", "iou"):
            leaves = [t.clone().requires_grad_(True) for t in cls + cnt + reg]
            res = ref_loss.FCOSLoss(mode)([[leaves[:5], leaves[5:10], leaves[10:]], tg])
            res[3].backward()
            out[f"{name}_{mode}_losses"] = np.array([float(r.detach()) for r in res], np.float64)
            for i in range(5):
                out[f"{name}_{mode}_gcls{i}"] = leaves[i].grad.numpy()
                out[f"{name}_{mode}_gcnt{i}"] = leaves[5 + i].grad.numpy()
                out[f"{name}_{mode}_greg{i}"] = leaves[10 + i].grad.numpy()
        for i in range(5):
            out[f"{name}_cls{i}"] = cls[i].numpy(); out[f"{name}_cnt{i}"] = cnt[i].numpy(); out[f"{name}_reg{i}"] = reg[i].numpy()
    out["gt"] = gt.numpy(); out["labels"] = labels.numpy(); out["strides"] = np.array(strides)
    save("g67_targets_loss", **out)


# ------------------------------------------------------------------ G8 tiny HISFCOS FPN + head, tiny FCOS FPN + head
def g8():
    gen = torch.Generator().manual_seed(8)
    torch.manual_seed(8)
    fpn = ref_his.HalfInvertedStageFPN([32, 64, 128], 32).eval()
    head = ref_his.HISFCOSHead(32, 20, 0.01).eval()
    randomize_bn(fpn, gen); randomize_bn(head, gen)
    with torch.no_grad():
        for i, s in enumerate(head.scale_exp):
            s.scale.fill_(0.8 + 0.1 * i)
    # features of a 128x128 input: 16x16 / 8x8 / 4x4, max-pooled to 2x2 (P6) and 1x1 (P7)
    c3 = torch.randn(2, 32, 16, 16, generator=gen); c4 = torch.randn(2, 64, 8, 8, generator=gen); c5 = torch.randn(2, 128, 4, 4, generator=gen)
    with torch.no_grad():
        ps = fpn((c3, c4, c5))
        cls, cnt, reg = head(ps)
    out = {"c3": c3.numpy(), "c4": c4.numpy(), "c5": c5.numpy()}
    for k, v in sd_np(fpn).items():
        out["sd.fpn." + k] = v
    for k, v in sd_np(head).items():
        out["sd.head." + k] = v
    for i in range(5):
        out[f"p{i}"] = ps[i].numpy(); out[f"cls{i}"] = cls[i].numpy(); out[f"cnt{i}"] = cnt[i].numpy(); out[f"reg{i}"] = reg[i].numpy()
    save("g8_tiny_hisfcos", **out)

    torch.manual_seed(81)
    fpn = ref_fcos.FeaturePyramidNetwork([128, 64, 32], 32).eval()
    head = ref_fcos.HeadFCOS(32, 20, 0.01).eval()
    randomize_bn(head, gen)
    with torch.no_grad():
        for m in head.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.mul_(8.0)  # std 0.01 init makes every output ~bias; scale up so the towers matter
        ps = fpn((c3, c4, c5))
        cls, cnt, reg = head(ps)
    out = {"c3": c3.numpy(), "c4": c4.numpy(), "c5": c5.numpy()}
    for k, v in sd_np(fpn).items():
        out["sd.FPN." + k] = v
    for k, v in sd_np(head).items():
        out["sd.head." + k] = v
    for i in range(5):
        out[f"p{i}"] = ps[i].numpy(); out[f"cls{i}"] = cls[i].numpy(); out[f"cnt{i}"] = cnt[i].numpy(); out[f"reg{i}"] = reg[i].numpy()
    save("g8_tiny_fcos", **out)


# ------------------------------------------------------------------ G9 full-width single layers / blocks
def g9():
    """Full-width (256-channel, 80-class) reference modules on 20x20 maps: HisBlock (1x1, depthwise, SE, 3x3, dilated 3x3),
    HISFCOSHead over two levels (pointwise / depthwise / GroupNorm(32, 512) / 3x3 towers / GroupNorm(32, 256) / 80-, 1- and
    4-wide predictors / ScaleExp), SEBlock(128, 4), DepthWiseConv2d(512).  Weights and inputs come from tests/golden/lcg.py
    (regenerated on both sides, not stored); stored: every 7th output element + float64 sums."""
    sys.path.insert(0, HERE)
    import lcg
    from model.modules import modules as ref_mod
    out = {}

    def put(name, t):
        a = t.detach().numpy().reshape(-1)
        out[name + "_shape"] = np.array(t.shape)
        out[name + "_s7"] = a[::7].copy()
        out[name + "_sum"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])

    blk = ref_his.HisBlock(256, 4, 2).eval()
    lcg.fill_state(blk, 91)
    x = lcg.tensor((1, 256, 20, 20), 9101)
    with torch.no_grad():
        put("hisblock", blk(x))
    head = ref_his.HISFCOSHead(256, 80, 0.01).eval()
    lcg.fill_state(head, 92)
    feats = [lcg.tensor((1, 256, 20, 20), 9201), lcg.tensor((1, 256, 10, 10), 9202)]
    with torch.no_grad():
        cls, cnt, reg = head(feats)
    for i in range(2):
        put(f"head_cls{i}", cls[i]); put(f"head_cnt{i}", cnt[i]); put(f"head_reg{i}", reg[i])
    se = ref_mod.SEBlock(128, 4).eval()
    lcg.fill_state(se, 93)
    with torch.no_grad():
        put("se", se(lcg.tensor((2, 128, 20, 20), 9301)))
    dw = ref_mod.DepthWiseConv2d(512, 3).eval()
    lcg.fill_state(dw, 94)
    with torch.no_grad():
        put("dw", dw(lcg.tensor((1, 512, 20, 20), 9401)))
    pw = ref_mod.PointWiseConv(2048, 256).eval()
    lcg.fill_state(pw, 95)
    with torch.no_grad():
        put("pw", pw(lcg.tensor((1, 2048, 20, 20), 9501)))
    save("g9_full_width", **out)


# ------------------------------------------------------------------ G10 MNFCOS pieces that run as shipped
def g10():
    """MNFCOS (model/od/MNFcos.py:11-36, what config/main.yaml:2 selects): the parts of it whose forward runs as shipped --
    MNHeadFCOS (two MNBlock(f, f, 3, 2, 2) + 3x3 / GroupNorm / SiLU towers + 1x1 predictors + ScaleExp, MNFcos.py:259-297) on a tiny
    pyramid, and MNBlock with k = 3 at dilation 1 and 2 (modules.py:195-216).  MNBlock pads its dilated depthwise conv with `dilation`,
    which keeps the map size only for k = 3: the k = 5 / 7 blocks of LieghtWeightFeaturePyramid_old (MNFcos.py:233-235) make
    `torch.add(x, x1)` (modules.py:215) raise, recorded here as `fpn_raises` -- there is no reference output to store for them."""
    from model.modules import modules as ref_mod
    from model.od import MNFcos as ref_mn
    gen = torch.Generator().manual_seed(10)
    torch.manual_seed(10)
    head = ref_mn.MNHeadFCOS(32, 20, 0.01).eval()
    randomize_bn(head, gen)
    with torch.no_grad():
        for i, s in enumerate(head.scale_exp):
            s.scale.fill_(0.8 + 0.1 * i)
    feats = [torch.randn(2, 32, h, h, generator=gen) for h in (16, 8, 4, 2, 1)]
    with torch.no_grad():
        cls, cnt, reg = head(feats)
    out = {}
    for k, v in sd_np(head).items():
        out["sd.head." + k] = v
    for i in range(5):
        out[f"f{i}"] = feats[i].numpy(); out[f"cls{i}"] = cls[i].numpy(); out[f"cnt{i}"] = cnt[i].numpy(); out[f"reg{i}"] = reg[i].numpy()
    for name, k, d in (("mnb_k3d1", 3, 1), ("mnb_k3d2", 3, 2)):
        blk = ref_mod.MNBlock(32, 32, k, d, 2).eval()
        randomize_bn(blk, gen)
        x = torch.randn(2, 32, 9, 7, generator=gen)
        with torch.no_grad():
            y = blk(x)
        out[name + ".x"] = x.numpy(); out[name + ".y"] = y.numpy()
        for kk, v in sd_np(blk).items():
            out[name + ".sd." + kk] = v
    raises = 0
    try:
        fpn = ref_mn.LieghtWeightFeaturePyramid_old([128, 64, 32], 32).eval()
        with torch.no_grad():
            fpn((torch.randn(1, 32, 16, 16), torch.randn(1, 64, 8, 8), torch.randn(1, 128, 4, 4)))
    except RuntimeError:
        raises = 1
    out["fpn_raises"] = np.array([raises])
    save("g10_mnfcos_parts", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g3b", "g4", "g5", "g67", "g8", "g9", "g10"]
    for name in which:
        globals()[name]()
