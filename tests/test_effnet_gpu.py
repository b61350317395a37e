"""GPU parity of the EfficientNet (MBConv) trunk and of FCOS(efficientnet=True) — BASELINE Cfg5 — against the oracle
(oracle/effnet_ref.py: a restatement of efficientnet_pytorch 0.7.1, third-party, PARITY UNPINNED) and of the new layer
kernels against plain torch fp32 references of the same op.  Tolerance: 1e-4 abs + 1e-4 rel (north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import effnet_ref as E
from oracle import torch_ref as R
from pytorch_object_detection_amd import ops
from pytorch_object_detection_amd._lib import ACT_NONE, ACT_RELU, ACT_SILU, FdError, Segs
from pytorch_object_detection_amd.model.backbone.efficientnetv1 import EfficientNetV1
from pytorch_object_detection_amd.model.modules.head import ClipBoxes, FCOSHead
from pytorch_object_detection_amd.model.od import FCOS
from effnet_init import init_effnet
from test_model_gpu import assert_same_detections

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(atol=1e-4, rtol=1e-4)


def to_rows(x):
    B, C, H, W = x.shape
    return ops.Rows(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV))


def from_rows(r, B, H, W):
    return r.tensor().reshape(B, H, W, -1).permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("case", [
    # C, K, stride, (pad_before, pad_after), H, W
    (144, 3, 2, (0, 1), 26, 42), (192, 5, 2, (2, 2), 13, 21), (40, 3, 1, (1, 1), 17, 9), (288, 5, 1, (2, 2), 12, 20),
    (96, 5, 2, (1, 2), 16, 24), (576, 3, 2, (0, 1), 7, 5), (32, 7, 1, (3, 3), 9, 11), (24, 5, 1, (2, 2), 3, 3),
])
def test_dwconv2d_vs_torch(case):
    C, K, s, pad, H, W = case
    gen = torch.Generator().manual_seed(C + K)
    B = 2
    x = torch.randn(B, C, H, W, generator=gen)
    w = torch.randn(C, 1, K, K, generator=gen) / K
    sc, sf = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.1
    ref = F.conv2d(F.pad(x, (pad[0], pad[1], pad[0], pad[1])), w, None, s, 0, 1, C)
    ref = ref * sc[None, :, None, None] + sf[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    Ho, Wo = ref.shape[2:]
    y = ops.new_rows(B * Ho * Wo, C, DEV)
    ops.dwconv2d(to_rows(x), ops.pack_dwk_weight(w).to(DEV), y, B, H, W, K, s, pad[0], pad[0], Ho, Wo, sc.to(DEV), sf.to(DEV), ACT_SILU)
    np.testing.assert_allclose(from_rows(y, B, Ho, Wo).numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("case", [
    # Cin, mid, K, stride, (pad_before, pad_after), H, W    (EfficientNet-B3 stage 2: 24 -> 144 k3 s2, 32 -> 192 k3 s1; stage 3: 32 -> 192 k5 s2, 48 -> 288 k5 s1)
    (24, 144, 3, 2, (0, 1), 52, 84), (32, 192, 3, 1, (1, 1), 30, 45), (32, 192, 5, 2, (1, 2), 26, 42), (48, 288, 5, 1, (2, 2), 13, 25),
    (16, 20, 3, 1, (1, 1), 5, 7), (40, 100, 5, 2, (2, 2), 9, 33), (8, 48, 3, 2, (0, 1), 64, 8),
])
def test_mbconv_expand_depthwise_fused_vs_torch(case):
    """fd_mbconv_expand_dw_nhwc: MBConvBlock's _expand_conv -> _bn0 -> swish -> _depthwise_conv (static SAME padding) -> _bn1 -> swish (efficientnet_pytorch 0.7.1
    behind model/backbone/efficientnetv1.py:11-26) in one launch -- the expanded map stays in LDS -- plus the SE pooling's per-tile partial sums.  Against torch in fp64
    (the padding of the EXPANDED map is zero, not swish(bn0(0)): the kernel must zero the halo pixels outside the image), channel views with NaN neighbours, ragged tile
    edges, widths that are no multiple of 32; the gates fd_se_gate_from_pool derives from the partials against the pooling of the stored output."""
    Cin, mid, K, s, pad, H, W = case
    gen = torch.Generator().manual_seed(Cin + mid + K + s)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=gen)
    we = torch.randn(mid, Cin, 1, 1, generator=gen) / Cin ** 0.5
    wd = torch.randn(mid, 1, K, K, generator=gen) / K
    sc0, sf0 = torch.rand(mid, generator=gen) + 0.5, torch.randn(mid, generator=gen) * 0.3
    sc1, sf1 = torch.rand(mid, generator=gen) + 0.5, torch.randn(mid, generator=gen) * 0.1
    e = F.conv2d(x.double(), we.double()) * sc0.double()[None, :, None, None] + sf0.double()[None, :, None, None]
    e = e * torch.sigmoid(e)
    ref = F.conv2d(F.pad(e, (pad[0], pad[1], pad[0], pad[1])), wd.double(), None, s, 0, 1, mid) * sc1.double()[None, :, None, None] + sf1.double()[None, :, None, None]
    ref = (ref * torch.sigmoid(ref)).float()
    Ho, Wo = ref.shape[2:]
    xb = torch.full((B * H * W, Cin + 8), float("nan"), device=DEV)
    xb[:, 4:4 + Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin).to(DEV)
    yb = torch.full((B * Ho * Wo, mid + 8), float("nan"), device=DEV)
    pool, T = ops.mbconv_pool_buffer(B, Ho, Wo, mid, K, s, DEV)
    pool.fill_(float("nan"))
    assert ops.mbconv_fused_ok(Cin, mid, K, s)
    ops.mbconv_expand_dw(ops.Rows(xb, 4, Cin), ops.pack_mbconv_expand_weight(we).to(DEV), sc0.to(DEV), sf0.to(DEV), ops.pack_dwk_weight(wd).to(DEV), sc1.to(DEV), sf1.to(DEV),
                         ops.Rows(yb, 4, mid), pool, B, H, W, K, s, pad[0], pad[0], Ho, Wo)
    assert torch.isnan(yb[:, :4]).all() and torch.isnan(yb[:, 4 + mid:]).all(), "wrote outside its channel view"
    got = yb[:, 4:4 + mid].cpu().reshape(B, Ho, Wo, mid).permute(0, 3, 1, 2)
    assert not torch.isnan(got).any(), "an output pixel was never written"
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=3e-5, rtol=2e-5)
    # pooling partials: every (image, tile, channel) written; their sum = the sum of the stored output
    assert not torch.isnan(pool).any()
    psum = pool.view(B, T, mid).double().sum(1).cpu()
    np.testing.assert_allclose(psum.numpy(), got.double().sum((2, 3)).numpy(), rtol=1e-5, atol=1e-4)
    Cr = max(1, Cin // 4)
    w1, b1 = (torch.randn(Cr, mid, generator=gen) / mid ** 0.5).to(DEV), torch.randn(Cr, generator=gen).to(DEV)
    w2, b2 = (torch.randn(mid, Cr, generator=gen) / Cr ** 0.5).to(DEV), torch.randn(mid, generator=gen).to(DEV)
    ws = ops.se_workspace(B, Ho * Wo, mid, DEV)
    g1 = ops.se_gate_from_pool(pool, T, w1, b1, w2, b2, B, Ho * Wo, mid, Cr, ws).clone()
    g0 = ops.se_gate(ops.Rows(yb, 4, mid), w1, b1, w2, b2, B, Ho * Wo, Cr, ops.se_workspace(B, Ho * Wo, mid, DEV)).clone()
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for bad in ((64, mid, K, s), (Cin, mid, 7, s), (20, mid, K, s)):
        assert not ops.mbconv_fused_ok(*bad)


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 37, 51)])
def test_stem_conv3_vs_torch(shape):
    B, H, W = shape
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, 3, H, W, generator=gen)
    w = torch.randn(40, 3, 3, 3, generator=gen) / 5
    sc, sf = torch.rand(40, generator=gen) + 0.5, torch.randn(40, generator=gen) * 0.1
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, 2) * sc[None, :, None, None] + sf[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    Ho, Wo = ref.shape[2:]
    x4 = torch.empty(B * H * W, 4, device=DEV)
    ops.nchw3_to_nhwc4(x.to(DEV), x4)
    y = ops.new_rows(B * Ho * Wo, 40, DEV)
    ops.stem_conv3(x4, ops.pack_stem3_weight(w).to(DEV), y, B, H, W, 3, 2, 0, 0, Ho, Wo, sc.to(DEV), sf.to(DEV), ACT_SILU)
    np.testing.assert_allclose(from_rows(y, B, Ho, Wo).numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("case", [(40, 24, 1, 0), (24, 144, 1, 0), (136, 816, 1, 0), (816, 136, 1, 0), (232, 1392, 1, 0),
                                  (48, 64, 3, 1), (136, 64, 3, 1), (4, 8, 1, 0)])
def test_conv_input_width_not_multiple_of_32(case, prec):
    """EfficientNet widths (Cin % 4 == 0, not % 32): the loader masks the partial 32-channel chunk.  The input is a channel
    slice of a wider buffer whose neighbouring channels are NaN, so an unmasked read would poison the result."""
    Cin, Cout, k, pad = case
    gen = torch.Generator().manual_seed(Cin * 7 + Cout)
    B, H, W = 2, 9, 13
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=gen) * 0.1
    ref = F.conv2d(x, w, b, 1, pad)
    wide = torch.full((B * H * W, Cin + 8), float("nan"), device=DEV)
    wide[:, 4:4 + Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin).to(DEV)
    y = ops.new_rows(B * H * W, Cout, DEV)
    segs = Segs.make(B, [(H, W)])
    wp = (ops.pack_conv_weight_f16x3 if prec == "f16x3" else ops.pack_conv_weight)(w.to(DEV))
    ops.conv_call(ops.Rows(wide, 4, Cin), segs, wp, y, Cin=Cin, Cout=Cout, k=k, pad=pad, shift=b.to(DEV),
                  precision=1 if prec == "f16x3" else 0)()
    np.testing.assert_allclose(from_rows(y, B, H, W).numpy(), ref.numpy(), **TOL)


@pytest.mark.parametrize("C,Cr,HW", [(144, 6, (13, 21)), (2304, 96, (4, 6)), (40, 10, (20, 30)), (816, 34, (1, 1))])
def test_se_any_width(C, Cr, HW):
    gen = torch.Generator().manual_seed(C)
    B, (H, W) = 3, HW
    x = torch.randn(B, C, H, W, generator=gen)
    w1, b1 = torch.randn(Cr, C, generator=gen) / C ** 0.5, torch.randn(Cr, generator=gen) * .1
    w2, b2 = torch.randn(C, Cr, generator=gen) / Cr ** 0.5, torch.randn(C, generator=gen) * .1
    g = torch.sigmoid(F.silu(x.mean((2, 3)) @ w1.t() + b1) @ w2.t() + b2)
    ref = x * g[:, :, None, None]
    r = to_rows(x)
    ws = ops.se_workspace(B, H * W, C, DEV)
    ops.se_scale(r, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), r, B, H * W, Cr, ws)      # in place, as the MBConv plan runs it
    np.testing.assert_allclose(from_rows(r, B, H, W).numpy(), ref.numpy(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("number,shape", [(3, (1, 128, 192)), (0, (2, 96, 64)), (3, (2, 224, 160))])
def test_efficientnet_endpoints_vs_oracle(number, shape):
    """EfficientNetV1(n)(x) -> the five endpoints of efficientnetv1.py:24-26 against the oracle's restatement."""
    B, H, W = shape
    net = EfficientNetV1(number).eval()
    init_effnet(net, 11 + number)
    sd = {"backbone." + k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref = E.efficientnet_endpoints5(sd, x, number)
    net.to(DEV)
    out = net(x.to(DEV))
    assert len(out) == 5
    for i, (o, r) in enumerate(zip(out, ref)):
        assert tuple(o.shape) == tuple(r.shape)
        np.testing.assert_allclose(o.cpu().numpy(), r.numpy(), err_msg=f"reduction_{i + 1}", **TOL)
    with pytest.raises(FdError, match="inference-only"):
        net.train()(x.to(DEV))


def _fcos_b3(ncls, feature, seed):
    torch.manual_seed(seed)
    model = FCOS([384, 136, 48], ncls, feature, efficientnet=True, backbone_number=3).eval()
    init_effnet(model.backbone, seed + 1)
    gen = torch.Generator().manual_seed(seed + 2)
    for m in model.head.modules():
        if isinstance(m, torch.nn.GroupNorm):
            m.weight.data.copy_(torch.rand(m.num_channels, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_channels, generator=gen) * 0.1)
        if isinstance(m, torch.nn.Conv2d):
            with torch.no_grad():
                m.weight.mul_(8.0)
    return model


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_fcos_b3_mixed_aspect_832x1344_vs_oracle(prec):
    """BASELINE Cfg5: EfficientNet-B3 FCOS on a mixed-aspect batch padded to 832x1344 (an 800x1333 image and an 800x1067 one,
    resized / padded as dataset/voc.py:117-132,149-156 do: the second image's right part is normalised zero padding).
    All 15 head outputs within 1e-4 of the oracle; detections identical to the oracle post-process on the device's outputs."""
    model = _fcos_b3(80, 256, 20)
    model.conv_precision = prec
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 832, 1344, generator=gen)
    pad_val = torch.tensor([-0.485 / 0.229, -0.456 / 0.224, -0.406 / 0.225])[None, :, None, None]
    x[0:1, :, 800:, :] = pad_val
    x[0:1, :, :, 1333:] = pad_val
    x[1:2, :, 800:, :] = pad_val
    x[1:2, :, :, 1067:] = pad_val
    with torch.no_grad():
        ref = E.fcos_effnet_forward(sd, x, 3)
    model.to(DEV)
    xd = x.to(DEV)
    out = model(xd)
    assert [tuple(t.shape[2:]) for t in out[0]] == [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)]
    for name, o, r in zip(("cls", "cnt", "reg"), out, ref):
        for i in range(5):
            assert tuple(o[i].shape) == tuple(r[i].shape)
            np.testing.assert_allclose(o[i].cpu().numpy(), r[i].numpy(), err_msg=f"{name}{i}", **TOL)
    head = FCOSHead(0.05, 0.6, 1000, [8, 16, 32, 64, 128])
    s, c, b, counts = head.detect_padded(out)
    b = ClipBoxes()(xd, b)
    exp = R.fcos_detect([[t.cpu() for t in grp] for grp in out], [8, 16, 32, 64, 128], 0.05, 0.6, 1000, (832, 1344))
    for bi in range(2):
        n = int(counts[bi])
        assert n == len(exp[bi][0])
        assert_same_detections(s[bi, :n].cpu().numpy(), c[bi, :n].cpu().numpy(), b[bi, :n].cpu().numpy(), *exp[bi])


def test_fcos_b0_as_the_reference_constructs_it():
    """FCOS(..., efficientnet=True) with the reference's hard-coded B0 (Fcos.py:31-32): widths [320, 112, 40]."""
    torch.manual_seed(2)
    model = FCOS([320, 112, 40], 20, 64, efficientnet=True).eval()
    init_effnet(model.backbone, 3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randn(1, 3, 160, 128)
    with torch.no_grad():
        ref = E.fcos_effnet_forward(sd, x, 0)
    out = model.to(DEV)(x.to(DEV))
    for o, r in zip(out, ref):
        for i in range(5):
            np.testing.assert_allclose(o[i].cpu().numpy(), r[i].numpy(), **TOL)
    with pytest.raises(FdError, match="in_channel"):
        FCOS([2048, 1024, 512], 20, 64, efficientnet=True)


def test_collate_u8_mixed_aspect_bit_exact():
    """dataset/voc.py:128-132 (pad to the next multiple of 32) + :141-156 (pad to the batch maximum, THEN Normalize) on the
    device, from the resized uint8 images: bit-identical to ToTensor -> pad(0.) -> Normalize."""
    rng = np.random.default_rng(4)
    sizes = [(100, 83), (83, 100), (96, 96), (31, 127)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    Hc = max(h + 32 - h % 32 for h, _ in sizes)
    Wc = max(w + 32 - w % 32 for _, w in sizes)
    assert (Hc, Wc) == (128, 128)          # 96 -> 128: an aligned side still gets a full extra 32 (voc.py:128-129)
    canvas = np.zeros((len(imgs), Hc, Wc, 3), np.uint8)          # uint8 zeros normalise to (0 - mean) / std
    for i, im in enumerate(imgs):
        canvas[i, :im.shape[0], :im.shape[1]] = im
    expect = R.normalize_u8(canvas)
    out, keep = ops.collate_u8([torch.from_numpy(im).to(DEV) for im in imgs], Hc, Wc, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    got = out.cpu().numpy().reshape(len(imgs), Hc, Wc, 4)
    np.testing.assert_array_equal(got[..., :3], expect)
    assert (got[..., 3] == 0).all()
    # the collated batch IS the stem's input layout: a model fed with it equals the model fed with the NCHW float batch
    torch.manual_seed(1)
    from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
    model = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).eval().to(DEV)
    a = [t.clone() for t in model(torch.from_numpy(expect).permute(0, 3, 1, 2).contiguous().to(DEV))[0]]
    b = model.forward_images([torch.from_numpy(im).to(DEV) for im in imgs])[0]
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_cfg1_builder_voc_512_vs_oracle(tmp_path):
    """BASELINE Cfg1 through the drop-in boundary: Builder(load_config(main.yaml -> voc.yaml, HISFCOS)).model_build()
    (bulider.py:15-26, config/voc.yaml:34-51) on 1x3x512x512 against the oracle's CPU path (the reference's own
    CPU-runnable case), then FCOSHead + ClipBoxes as test.py:172-176,205-207 call them."""
    import os
    import pytorch_object_detection_amd as pkg
    from pytorch_object_detection_amd.bulider import Builder, load_config
    from test_model_gpu import randomize_norms
    cfg_dir = os.path.join(os.path.dirname(pkg.__file__), "config")
    main = tmp_path / "config" / "main.yaml"
    main.parent.mkdir()
    main.write_text(open(os.path.join(cfg_dir, "main.yaml")).read().replace("dataset: COCO", "dataset: VOC")
                    .replace("config/voc.yaml", os.path.join(cfg_dir, "voc.yaml")))
    cfg = load_config(str(main))
    assert cfg["model"]["name"] == "HISFCOS" and cfg["dataset_setting"]["class_num"] == 20 and cfg["dataset_setting"]["input"] == [512, 512]
    torch.manual_seed(0)
    model = Builder(cfg).model_build().eval()
    randomize_norms(model, 4)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randn(1, 3, 512, 512)
    with torch.no_grad():
        ref = R.hisfcos_forward(sd, x)
    model.to(DEV)
    xd = x.to(DEV)
    out = model(xd)
    assert [tuple(t.shape[2:]) for t in out[0]] == [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]      # sum HW = 5456
    for name, o, r in zip(("cls", "cnt", "reg"), out, ref):
        for i in range(5):
            np.testing.assert_allclose(o[i].cpu().numpy(), r[i].numpy(), err_msg=f"{name}{i}", **TOL)
    strides = cfg["HISFCOS"]["stride"]
    scores, classes, boxes = FCOSHead(0.05, 0.6, 1000, strides)(out)          # batch 1: the reference's stacked return
    boxes = ClipBoxes()(xd, boxes.contiguous())
    (es, ec, eb), = R.fcos_detect([[t.cpu() for t in g] for g in out], strides, 0.05, 0.6, 1000, (512, 512))
    assert_same_detections(scores[0].cpu().numpy(), classes[0].cpu().numpy(), boxes[0].cpu().numpy(), es, ec, eb)
