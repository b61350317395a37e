"""GPU parity of the layer kernels (C-ABI) against torch fp32 CPU references of the same op.
Tolerance for conv / norm outputs: 1e-4 absolute + 1e-5 relative (north_star: 'conv ... within 1e-4 fp32')."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pytorch_object_detection_amd import ops
from pytorch_object_detection_amd._lib import ACT_EXP, ACT_NONE, ACT_RELU, ACT_SILU, Segs

pytestmark = pytest.mark.gpu
from pytorch_object_detection_amd._lib import TILES as _TILE_IDS  # noqa: E402
DEV = "cuda:0"
ATOL, RTOL = 1e-4, 1e-5


def to_rows(x):  # NCHW cpu -> Rows on device
    B, C, H, W = x.shape
    return ops.Rows(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV))


def from_rows(r, B, H, W):
    return r.tensor().reshape(B, H, W, -1).permute(0, 3, 1, 2).cpu()


def act_ref(v, act):
    return {ACT_NONE: lambda t: t, ACT_RELU: F.relu, ACT_SILU: F.silu}[act](v)


CONV_CASES = [
    # Cin, Cout, k, stride, pad, dil, H, W, act, res, bn
    (64, 64, 1, 1, 0, 1, 20, 20, ACT_RELU, False, True),
    (64, 256, 1, 1, 0, 1, 13, 21, ACT_NONE, True, True),
    (256, 128, 1, 2, 0, 1, 20, 20, ACT_NONE, False, True),        # downsample 1x1 s2
    (128, 128, 3, 2, 1, 1, 21, 13, ACT_RELU, False, True),        # 3x3 s2 on odd sizes
    (256, 256, 3, 1, 2, 2, 20, 20, ACT_SILU, False, True),        # HisBlock conv4: dilation 2
    (256, 128, 3, 1, 1, 1, 5, 5, ACT_RELU, False, True),          # tiny level
    (256, 80, 3, 1, 1, 1, 10, 10, ACT_NONE, False, False),        # cls_logits (bias only)
    (256, 5, 3, 1, 1, 1, 10, 10, ACT_NONE, False, False),         # cnt+reg
    (2048, 256, 1, 1, 0, 1, 20, 20, ACT_RELU, False, True),       # FPN lateral, K=2048
    (32, 32, 3, 1, 1, 1, 16, 16, ACT_NONE, False, False),
]


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_single_level(case, prec):
    Cin, Cout, k, stride, pad, dil, H, W, act, use_res, use_bn = case
    gen = torch.Generator().manual_seed(hash(case) % 1000)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    scale = torch.rand(Cout, generator=gen) + 0.5 if use_bn else None
    shift = torch.randn(Cout, generator=gen)
    ref = F.conv2d(x, w, None, stride, pad, dil)
    ref = ref * (scale.view(1, -1, 1, 1) if use_bn else 1.0) + shift.view(1, -1, 1, 1)
    Ho, Wo = ref.shape[2:]
    res = torch.randn(B, Cout, Ho, Wo, generator=gen) if use_res else None
    if use_res:
        ref = ref + res
    ref = act_ref(ref, act)
    segs = Segs.make(B, [(H, W)])
    xr = to_rows(x)
    y = ops.new_rows(B * Ho * Wo, Cout, DEV)
    rr = to_rows(res) if use_res else None
    wp = ops.pack_conv_weight_f16x3(w.to(DEV)) if prec == "f16x3" else ops.pack_conv_weight(w.to(DEV))
    from pytorch_object_detection_amd._lib import PATCH_TILE, WAVE_TILE
    patch_ok = k == 3 and stride == 1 and pad == dil
    wave_ok = prec == "f32" and ops.wave_ok(Cin, Cout, k, stride, pad)
    wf = ops.pack_conv_weight_wave(w.to(DEV)) if wave_ok else None
    for tile in ([0] if prec == "f32" else []) + sorted(_TILE_IDS):
        y.buf.fill_(float("nan"))
        call = ops.conv_call(xr, segs, wp, y, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil,
                             scale=scale.to(DEV) if use_bn else None, shift=shift.to(DEV), res=rr, act=act, tile=tile,
                             precision=1 if prec == "f16x3" else 0, w_frag=wf)
        if tile == PATCH_TILE and not patch_ok:      # the patch tile is 3x3 stride-1 'same' only: a clean error, no launch
            with pytest.raises(Exception, match="PATCH"):
                call()
            continue
        if tile == WAVE_TILE and not wave_ok:        # the wave tile is fp32 1x1 stride-1 with Cin, Cout multiples of 32 and its own weight packing
            with pytest.raises(Exception, match="WAVE64"):
                call()
            continue
        call()
        np.testing.assert_allclose(from_rows(y, B, Ho, Wo).numpy(), ref.numpy(), atol=ATOL, rtol=RTOL, err_msg=f"tile {tile}")


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("ksplit", [2, 3, 8])
def test_conv_split_k(ksplit, prec):
    """K split over several workgroups per tile + deterministic combine launch (epilogue incl. residual / act / odd Cout)."""
    gen = torch.Generator().manual_seed(21 + ksplit)
    B, Cin, H, W = 2, 256, 10, 10
    for Cout, act, use_res in ((128, ACT_RELU, True), (5, ACT_NONE, False)):
        x = torch.randn(B, Cin, H, W, generator=gen)
        w = torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9)
        scale, shift = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen)
        res = torch.randn(B, Cout, H, W, generator=gen) if use_res else None
        ref = F.conv2d(x, w, None, 1, 2, 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        ref = act_ref(ref + res if use_res else ref, act)
        segs = Segs.make(B, [(H, W)])
        y = ops.new_rows(B * H * W, 8 if Cout == 5 else Cout, DEV)
        yv = ops.Rows(y.buf, 0, Cout)
        ws = torch.empty(ksplit * B * H * W * ((Cout + 3) & ~3), device=DEV)
        wp = ops.pack_conv_weight_f16x3(w.to(DEV)) if prec == "f16x3" else ops.pack_conv_weight(w.to(DEV))
        outs = []
        for _ in range(2):
            y.buf.fill_(float("nan"))
            ops.conv_call(to_rows(x), segs, wp, yv, Cin=Cin, Cout=Cout, k=3, pad=2, dil=2, scale=scale.to(DEV),
                          shift=shift.to(DEV), res=to_rows(res) if use_res else None, act=act, tile=4, ksplit=ksplit,
                          workspace=ws, precision=1 if prec == "f16x3" else 0)()
            outs.append(yv.tensor().clone())
        assert torch.equal(outs[0], outs[1])                     # fixed combine order: bitwise reproducible
        got = outs[0].reshape(B, H, W, Cout).permute(0, 3, 1, 2).cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=ATOL, rtol=RTOL)


def test_conv_stem_7x7():
    gen = torch.Generator().manual_seed(7)
    B, H, W = 2, 64, 96
    x = torch.randn(B, 3, H, W, generator=gen)
    w = torch.randn(64, 3, 7, 7, generator=gen) / np.sqrt(147)
    scale, shift = torch.rand(64, generator=gen) + 0.5, torch.randn(64, generator=gen)
    ref = F.relu(F.conv2d(x, w, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    x4 = torch.empty(B * H * W, 4, device=DEV)
    ops.nchw3_to_nhwc4(x.to(DEV), x4)
    assert torch.equal(x4.view(B, H, W, 4)[..., :3].permute(0, 3, 1, 2).cpu(), x) and x4[:, 3].abs().sum() == 0
    y = ops.new_rows(B * 32 * 48, 64, DEV)
    ops.conv_call(ops.Rows(x4), Segs.make(B, [(H, W)]), ops.pack_stem_weight(w.to(DEV)), y, Cin=4, Cout=64, k=7, stride=2,
                  pad=3, scale=scale.to(DEV), shift=shift.to(DEV), act=ACT_RELU, stem=True)()
    np.testing.assert_allclose(from_rows(y, B, 32, 48).numpy(), ref.numpy(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("hw", [(64, 96), (70, 50), (32, 32), (130, 258)])
def test_stem7x7_kernel(hw):
    """fd_stem7x7_nhwc4 (LDS-staged patch + filter bank, K = 7 x 22) against F.conv2d: full and partial 8 x 32 output tiles, BN + ReLU,
    output written into a channel slice."""
    gen = torch.Generator().manual_seed(hw[0] + hw[1])
    B, (H, W) = 2, hw
    x = torch.randn(B, 3, H, W, generator=gen)
    w = torch.randn(64, 3, 7, 7, generator=gen) / np.sqrt(147)
    scale, shift = torch.rand(64, generator=gen) + 0.5, torch.randn(64, generator=gen)
    ref = F.relu(F.conv2d(x, w, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    Ho, Wo = ref.shape[2:]
    x4 = torch.empty(B * H * W, 4, device=DEV)
    ops.nchw3_to_nhwc4(x.to(DEV), x4)
    yb = torch.full((B * Ho * Wo, 72), float("nan"), device=DEV)
    ops.stem7x7(ops.Rows(x4), ops.pack_stem7_weight(w.to(DEV)), ops.Rows(yb, 4, 64), B, H, W, scale.to(DEV), shift.to(DEV), ACT_RELU)
    got = yb.cpu()
    assert torch.isnan(got[:, :4]).all() and torch.isnan(got[:, 68:]).all()
    np.testing.assert_allclose(got[:, 4:68].reshape(B, Ho, Wo, 64).permute(0, 3, 1, 2).numpy(), ref.numpy(), atol=ATOL, rtol=RTOL)


def test_conv_pyramid_channel_views_and_exp():
    """5 levels in one launch, input read from a channel slice, output written into a channel slice,
    exp(scale_level * x) on channels >= 1 (the fused cnt_logits + reg_pred conv)."""
    gen = torch.Generator().manual_seed(3)
    B, F_ = 2, 64
    hw = [(12, 20), (6, 10), (3, 5), (2, 3), (1, 1)]
    segs = Segs.make(B, hw)
    feats = [torch.randn(B, 2 * F_, h, w, generator=gen) for h, w in hw]
    w = torch.randn(5, F_, 3, 3, generator=gen) / np.sqrt(9 * F_)
    bias = torch.randn(5, generator=gen) * 0.1
    scales = [0.8, 0.9, 1.0, 1.1, 1.2]
    buf = torch.cat([f.permute(0, 2, 3, 1).reshape(-1, 2 * F_) for f in feats]).contiguous().to(DEV)
    out = torch.full((segs.rows, 8), -7.0, device=DEV)
    ops.conv_call(ops.Rows(buf, F_, F_), segs, ops.pack_conv_weight(w.to(DEV)), ops.Rows(out, 2, 5), Cin=F_, Cout=5, k=3,
                  pad=1, shift=bias.to(DEV), act=ACT_EXP, act_c0=1, seg_param=scales)()
    o = out.cpu()
    assert (o[:, :2] == -7).all() and (o[:, 7] == -7).all()
    for i, (f, s) in enumerate(zip(feats, scales)):
        ref = F.conv2d(f[:, F_:], w, bias, 1, 1)
        ref = torch.cat([ref[:, :1], torch.exp(ref[:, 1:] * s)], 1)
        got = o[segs.m_start[i]:segs.m_start[i + 1], 2:7].reshape(B, hw[i][0], hw[i][1], 5).permute(0, 3, 1, 2)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=ATOL, rtol=1e-4)


def test_conv_full_width_head_tower_sample():
    """The K6 shape at full width (256 -> 512, 3x3, K=2304) on a 40x40 + 20x20 pyramid."""
    gen = torch.Generator().manual_seed(9)
    B = 1
    hw = [(40, 40), (20, 20)]
    segs = Segs.make(B, hw)
    feats = [torch.randn(B, 256, h, w, generator=gen) for h, w in hw]
    w = torch.randn(512, 256, 3, 3, generator=gen) / np.sqrt(2304)
    buf = torch.cat([f.permute(0, 2, 3, 1).reshape(-1, 256) for f in feats]).contiguous().to(DEV)
    y = ops.new_rows(segs.rows, 512, DEV)
    ops.conv_call(ops.Rows(buf), segs, ops.pack_conv_weight(w.to(DEV)), y, Cin=256, Cout=512, k=3, pad=1)()
    o = y.buf.cpu()
    for i, f in enumerate(feats):
        ref = F.conv2d(f, w, None, 1, 1)
        got = o[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, hw[i][0], hw[i][1], 512).permute(0, 3, 1, 2)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("k,s,pad,H,W", [(3, 2, 1, 32, 48), (2, 2, 0, 20, 20), (2, 2, 0, 13, 21)])
def test_maxpool_and_add(k, s, pad, H, W):
    gen = torch.Generator().manual_seed(1)
    B, C = 2, 64
    x = torch.randn(B, C, H, W, generator=gen)
    ref = F.max_pool2d(x, k, s, pad)
    Ho, Wo = ref.shape[2:]
    add = torch.randn(B, C, Ho, Wo, generator=gen)
    y = ops.new_rows(B * Ho * Wo, C, DEV)
    ops.maxpool(to_rows(x), y, B, H, W, k, s, pad)
    assert torch.equal(from_rows(y, B, Ho, Wo), ref)
    ops.maxpool(to_rows(x), y, B, H, W, k, s, pad, add=to_rows(add))
    assert torch.equal(from_rows(y, B, Ho, Wo), ref + add)


def test_upsample2x_add():
    gen = torch.Generator().manual_seed(2)
    B, C, H, W = 2, 32, 5, 7
    x, lat = torch.randn(B, C, H, W, generator=gen), torch.randn(B, C, 2 * H, 2 * W, generator=gen)
    y = ops.new_rows(B * 4 * H * W, C, DEV)
    ops.upsample2x_add(to_rows(x), to_rows(lat), y, B, H, W)
    assert torch.equal(from_rows(y, B, 2 * H, 2 * W), F.interpolate(x, scale_factor=2.0, mode="nearest") + lat)


def test_dwconv3x3_pyramid():
    gen = torch.Generator().manual_seed(4)
    B, C = 2, 128
    hw = [(9, 11), (4, 5), (1, 1)]
    segs = Segs.make(B, hw)
    feats = [torch.randn(B, C, h, w, generator=gen) for h, w in hw]
    w = torch.randn(C, 1, 3, 3, generator=gen) / 3
    scale, shift = torch.rand(C, generator=gen) + .5, torch.randn(C, generator=gen)
    buf = torch.cat([f.permute(0, 2, 3, 1).reshape(-1, C) for f in feats]).contiguous().to(DEV)
    y = ops.new_rows(segs.rows, C, DEV)
    ops.dwconv3x3(ops.Rows(buf), ops.pack_dw_weight(w.to(DEV)), y, segs, scale.to(DEV), shift.to(DEV), ACT_RELU)
    o = y.buf.cpu()
    for i, f in enumerate(feats):
        ref = F.relu(F.conv2d(f, w, None, 1, 1, 1, C) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        got = o[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, hw[i][0], hw[i][1], C).permute(0, 3, 1, 2)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("C,G,act", [(512, 32, ACT_RELU), (512, 64, ACT_SILU), (256, 32, ACT_RELU), (32, 32, ACT_NONE)])
def test_groupnorm_pyramid(C, G, act):
    gen = torch.Generator().manual_seed(C + G)
    B = 3
    hw = [(80, 80), (10, 10), (5, 5), (1, 1)]
    segs = Segs.make(B, hw)
    feats = [torch.randn(B, C, h, w, generator=gen) * 2 + 0.5 for h, w in hw]
    gamma, beta = torch.rand(C, generator=gen) + .5, torch.randn(C, generator=gen)
    buf = torch.cat([f.permute(0, 2, 3, 1).reshape(-1, C) for f in feats]).contiguous().to(DEV)
    x = ops.Rows(buf)
    ws = ops.groupnorm_workspace(segs, G, DEV)
    ops.groupnorm_act(x, gamma.to(DEV), beta.to(DEV), x, segs, G, act, ws)
    o = buf.cpu()
    for i, f in enumerate(feats):
        if hw[i] == (1, 1) and C // G == 1:
            continue  # a single-element group has zero variance: output is beta either way, checked below
        ref = act_ref(F.group_norm(f, G, gamma, beta, 1e-5), act)
        got = o[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, hw[i][0], hw[i][1], C).permute(0, 3, 1, 2)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


def test_se_block():
    gen = torch.Generator().manual_seed(6)
    B, C, Cr, H, W = 3, 128, 32, 20, 20
    x = torch.randn(B, C, H, W, generator=gen)
    w1, b1 = torch.randn(Cr, C, generator=gen) / 8, torch.randn(Cr, generator=gen) * .1
    w2, b2 = torch.randn(C, Cr, generator=gen) / 4, torch.randn(C, generator=gen) * .1
    g = torch.sigmoid(F.silu(x.mean((2, 3)) @ w1.t() + b1) @ w2.t() + b2)
    ref = x * g[:, :, None, None]
    out = torch.zeros(B * H * W, 2 * C, device=DEV)
    ws = ops.se_workspace(B, H * W, C, DEV)
    ops.se_scale(to_rows(x), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), ops.Rows(out, C, C), B, H * W, Cr, ws)
    got = out[:, C:].reshape(B, H, W, C).permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-5, rtol=1e-5)
    assert out[:, :C].abs().sum() == 0


def test_preprocess_u8_and_box_rescale_bit_exact():
    from oracle import torch_ref as R
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (2, 64, 96, 3), dtype=np.uint8)
    img[:, 50:, :, :] = 0                                    # zero padding rows normalise to -mean/std, as in the reference
    y = torch.empty(2 * 64 * 96, 4, device=DEV)
    ops.preprocess_u8(torch.from_numpy(img).to(DEV), y, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    got = y.cpu().numpy().reshape(2, 64, 96, 4)
    np.testing.assert_array_equal(got[..., :3], R.normalize_u8(img))
    assert (got[..., 3] == 0).all()
    boxes = rng.uniform(0, 600, (3, 40, 4)).astype(np.float32)
    out = ops.boxes_rescale_xywh_(torch.from_numpy(boxes).to(DEV), 1.7)
    np.testing.assert_array_equal(out.cpu().numpy(), R.boxes_rescale_xywh(boxes, 1.7))


WGRAD_CASES = [
    # Cin, Cout, k, stride, pad, dil, H, W
    (64, 64, 1, 1, 0, 1, 12, 20),
    (128, 256, 3, 1, 1, 1, 10, 14),
    (256, 256, 3, 1, 2, 2, 9, 9),
    (64, 128, 3, 2, 1, 1, 13, 21),
    (256, 128, 1, 2, 0, 1, 10, 10),
    (256, 80, 3, 1, 1, 1, 6, 6),
    (512, 256, 1, 1, 0, 1, 5, 5),
    (256, 32, 3, 1, 1, 1, 11, 9),     # narrow predictors: 32-channel tile variant
    (128, 20, 3, 1, 1, 1, 7, 12),
    (128, 8, 1, 1, 0, 1, 9, 9),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_backward_weight_and_data(case):
    """Weight gradient kernel (any stride) and data gradient through the forward kernel (stride 1) vs torch autograd."""
    Cin, Cout, k, stride, pad, dil, H, W = case
    gen = torch.Generator().manual_seed(sum(case))
    B = 3
    x = torch.randn(B, Cin, H, W, generator=gen, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad, dil)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    segs = Segs.make(B, [(H, W)])
    dw = ops.conv_wgrad(to_rows(x.detach()), to_rows(dy), segs, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil)
    ref_dw = w.grad.permute(0, 2, 3, 1)
    scale = float(ref_dw.abs().max())
    np.testing.assert_allclose(dw.cpu().numpy() / scale, ref_dw.numpy() / scale, atol=2e-5)
    if stride == 1 and Cout % 32 == 0:
        dx = ops.new_rows(B * H * W, Cin, DEV)
        ops.conv_call(to_rows(dy), Segs.make(B, [(Ho, Wo)]), ops.dgrad_weight(w.to(DEV)), dx, Cin=Cout, Cout=Cin, k=k,
                      stride=1, pad=dil * (k - 1) - pad, dil=dil)()
        scale = float(x.grad.abs().max())
        np.testing.assert_allclose(from_rows(dx, B, H, W).numpy() / scale, x.grad.numpy() / scale, atol=2e-5)


def test_conv_backward_weight_pyramid():
    gen = torch.Generator().manual_seed(77)
    B, Cin, Cout = 2, 256, 128
    hw = [(12, 20), (6, 10), (3, 5), (1, 1)]
    segs = Segs.make(B, hw)
    xs = [torch.randn(B, Cin, h, w, generator=gen) for h, w in hw]
    wt = (torch.randn(Cout, Cin, 3, 3, generator=gen) / 48).requires_grad_(True)
    dys = [torch.randn(B, Cout, h, w, generator=gen) for h, w in hw]
    sum((F.conv2d(x, wt, None, 1, 1) * dy).sum() for x, dy in zip(xs, dys)).backward()
    xb = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).contiguous().to(DEV)
    db = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in dys]).contiguous().to(DEV)
    dw = ops.conv_wgrad(ops.Rows(xb), ops.Rows(db), segs, Cin=Cin, Cout=Cout, k=3, pad=1)
    ref = wt.grad.permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dw.cpu().numpy() / scale, ref.numpy() / scale, atol=2e-5)


@pytest.mark.parametrize("C", [128, 512, 1024])
def test_dwconv3x3_backward_weight_pyramid(C):
    """Depthwise 3x3 weight gradient over a pyramid vs torch autograd (CPU fp32 reference)."""
    gen = torch.Generator().manual_seed(C)
    B = 2
    hw = [(9, 14), (5, 7), (2, 3), (1, 1)]
    segs = Segs.make(B, hw)
    xs = [torch.randn(B, C, h, w, generator=gen) for h, w in hw]
    wt = torch.randn(C, 1, 3, 3, generator=gen).requires_grad_(True)
    dys = [torch.randn(B, C, h, w, generator=gen) for h, w in hw]
    sum((F.conv2d(x, wt, None, 1, 1, 1, C) * dy).sum() for x, dy in zip(xs, dys)).backward()
    xb = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, C) for t in xs]).contiguous().to(DEV)
    db = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, C) for t in dys]).contiguous().to(DEV)
    dw = ops.dwconv3x3_wgrad(ops.Rows(xb), ops.Rows(db), segs)            # [9][C]
    ref = wt.grad.reshape(C, 9).t()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dw.cpu().numpy() / scale, ref.numpy() / scale, atol=2e-5)


def _frozen_bn(C, gen):
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=gen) * 0.2)
        bn.running_mean.copy_(torch.randn(C, generator=gen) * 0.3)
        bn.running_var.copy_(torch.rand(C, generator=gen) + 0.5)
    bn.eval()
    for p in bn.parameters():
        p.requires_grad = False
    return bn


@pytest.mark.parametrize("case", [
    # Cin, Cout, k, stride, dil, bias, groups, act, residual
    (64, 96, 3, 1, 1, False, 1, "relu", True),
    (64, 64, 1, 1, 1, True, 1, "relu", False),
    (128, 64, 3, 2, 1, False, 1, "relu", False),     # strided: stock data gradient, HIP weight gradient
    (64, 64, 3, 1, 2, False, 1, "none", True),
    (128, 128, 3, 1, 1, False, 128, "relu", False),  # depthwise
    (128, 128, 3, 1, 1, False, 128, "none", False),
])
def test_fused_conv_bn_act_autograd(case):
    """train_ops.conv_bn_act (one fused HIP launch + HIP backward kernels) vs the stock module chain on the CPU."""
    from pytorch_object_detection_amd import train_ops
    Cin, Cout, k, stride, dil, bias, groups, act, use_res = case
    gen = torch.Generator().manual_seed(sum(c if isinstance(c, int) else 0 for c in case))
    B, H, W = 2, 12, 10
    conv = torch.nn.Conv2d(Cin, Cout, k, stride, dil * (k - 1) // 2, dil, groups, bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) / np.sqrt(Cin // groups * k * k))
    bn = _frozen_bn(Cout, gen)
    x = torch.randn(B, Cin, H, W, generator=gen)
    Ho = (H + 2 * (dil * (k - 1) // 2) - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * (dil * (k - 1) // 2) - dil * (k - 1) - 1) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=gen) if use_res else None
    dy = torch.randn(B, Cout, Ho, Wo, generator=gen)

    def run(dev, fused):
        c, b = copy.deepcopy(conv).to(dev), copy.deepcopy(bn).to(dev)
        xx = x.detach().clone().to(dev).requires_grad_(True)
        rr = res.detach().clone().to(dev).requires_grad_(True) if use_res else None
        if fused:
            y = train_ops.conv_bn_act(c, b, xx, ops.ACT_RELU if act == "relu" else ops.ACT_NONE, rr)
        else:
            y = b(c(xx))
            if rr is not None:
                y = y + rr
            y = F.relu(y) if act == "relu" else y
        y.backward(dy.to(dev))
        grads = [xx.grad, c.weight.grad] + ([c.bias.grad] if bias else []) + ([rr.grad] if use_res else [])
        return y.detach().cpu(), [g.cpu() for g in grads], y

    y_ref, g_ref, _ = run("cpu", False)
    y_hip, g_hip, y_node = run(DEV, True)
    names, stack = set(), [y_node.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is not None and fn not in names:
            names.add(fn)
            stack.extend(f for f, _ in fn.next_functions)
    assert any(type(fn).__name__ in ("_ConvRowsBackward", "_DwRowsBackward") for fn in names)
    np.testing.assert_allclose(y_hip.numpy(), y_ref.numpy(), atol=2e-5, rtol=1e-5)
    for a, b_ in zip(g_hip, g_ref):
        scale = float(b_.abs().max())
        np.testing.assert_allclose(a.numpy() / scale, b_.numpy() / scale, atol=3e-5)
    assert train_ops.STATS["cl_copies"] >= 0


@pytest.mark.parametrize("act", ["none", "relu", "silu"])
@pytest.mark.parametrize("C,G", [(256, 32), (512, 32), (64, 8)])
def test_groupnorm_rows_autograd(C, G, act):
    """train_ops.groupnorm_rows (HIP forward + backward over a pyramid) vs nn.GroupNorm + activation per level on the CPU."""
    from pytorch_object_detection_amd import train_ops
    gen = torch.Generator().manual_seed(C + G + len(act))
    B = 3
    hw = [(9, 13), (5, 7), (3, 4), (1, 2), (1, 1)]
    gn = torch.nn.GroupNorm(G, C)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(C, generator=gen) + 0.5)
        gn.bias.copy_(torch.randn(C, generator=gen) * 0.3)
    fn = {"none": lambda t: t, "relu": F.relu, "silu": F.silu}[act]
    xs = [(torch.randn(B, C, h, w, generator=gen) * 1.5 + 0.3).requires_grad_(True) for h, w in hw]
    dys = [torch.randn(B, C, h, w, generator=gen) for h, w in hw]
    ys = [fn(gn(x)) for x in xs]
    sum((y * dy).sum() for y, dy in zip(ys, dys)).backward()
    g2 = copy.deepcopy(gn).to(DEV)
    g2.weight.grad = g2.bias.grad = None
    xd = [x.detach().clone().to(DEV).to(memory_format=torch.channels_last).requires_grad_(True) for x in xs]
    rows, segs = train_ops.pyramid_rows(xd)
    out = train_ops.groupnorm_rows(g2, rows, segs, {"none": ACT_NONE, "relu": ACT_RELU, "silu": ACT_SILU}[act])
    outs = train_ops.pyramid_split(out, segs)
    sum((y * dy.to(DEV)).sum() for y, dy in zip(outs, dys)).backward()
    for a, b in zip(outs, ys):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), atol=3e-5, rtol=1e-5)
    for a, b in zip(xd, xs):
        scale = float(b.grad.abs().max())
        np.testing.assert_allclose(a.grad.cpu().numpy() / scale, b.grad.numpy() / scale, atol=5e-5)
    for a, b in ((g2.weight.grad, gn.weight.grad), (g2.bias.grad, gn.bias.grad)):
        scale = float(b.abs().max())
        np.testing.assert_allclose(a.cpu().numpy() / scale, b.numpy() / scale, atol=5e-5)


@pytest.mark.parametrize("shape", [(64, 96, 3), (256, 64, 1), (32, 160, 5)])
def test_pack_conv_weight_kernel(shape):
    """fd_pack_conv_weight_f32 == the torch packers, bit for bit (forward layout and flipped / transposed / scaled dgrad layout)."""
    Cout, Cin, k = shape
    gen = torch.Generator().manual_seed(Cout + Cin + k)
    w = torch.randn(Cout, Cin, k, k, generator=gen).to(DEV)
    sc = (torch.rand(Cout, generator=gen) + 0.5).to(DEV)
    if Cin % 32 == 0:
        assert torch.equal(ops.pack_conv_weight_hip(w), ops.pack_conv_weight(w))
    if Cout % 32 == 0:
        assert torch.equal(ops.pack_conv_weight_hip(w, dgrad=True), ops.dgrad_weight(w))
        assert torch.equal(ops.pack_conv_weight_hip(w, sc, dgrad=True), ops.dgrad_weight(w * sc.view(-1, 1, 1, 1)))


def test_conv_wgrad_oihw_scaled():
    gen = torch.Generator().manual_seed(5)
    B, Cin, Cout, H, W = 2, 64, 96, 9, 7
    x = ops.Rows(torch.randn(B * H * W, Cin, generator=gen).to(DEV))
    dy = ops.Rows(torch.randn(B * H * W, Cout, generator=gen).to(DEV))
    sc = (torch.rand(Cout, generator=gen) + 0.5).to(DEV)
    segs = Segs.make(B, [(H, W)])
    ref = ops.conv_wgrad(x, dy, segs, Cin=Cin, Cout=Cout, k=3, pad=1)                       # OHWI
    got = ops.conv_wgrad(x, dy, segs, Cin=Cin, Cout=Cout, k=3, pad=1, scale=sc, oihw=True)
    assert got.shape == (Cout, Cin, 3, 3) and got.is_contiguous()
    np.testing.assert_allclose(got.cpu().numpy(), (ref * sc.view(-1, 1, 1, 1)).permute(0, 3, 1, 2).cpu().numpy(), rtol=1e-6, atol=1e-6)
    d0 = ops.dwconv3x3_wgrad(x, ops.Rows(dy.buf[:, :Cin].contiguous()), segs)
    d1 = ops.dwconv3x3_wgrad(x, ops.Rows(dy.buf[:, :Cin].contiguous()), segs, sc[:Cin].contiguous(), torch_layout=True)
    np.testing.assert_allclose(d1.cpu().numpy(), (d0 * sc[:Cin]).t().reshape(Cin, 1, 3, 3).cpu().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", [
    # Cin, Cout, k, stride, dil, levels (the train step's real shapes: batch 16, 512x512 input)
    (256, 256, 3, 1, 1, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]),     # head tower over the pyramid
    (256, 256, 3, 1, 2, [(64, 64)]),                                          # HisBlock conv4 (dilated) at P3
    (1024, 256, 1, 1, 1, [(32, 32)]),                                         # layer3 bottleneck 1x1
    (128, 128, 3, 2, 1, [(128, 128)]),                                        # layer2 strided 3x3 (weight gradient only)
])
def test_conv_backward_adjoint_identity_full_size(case):
    """Size-independent property at the bench sizes (no oracle needed): the conv is linear in x and in w, so
    <conv(x, w), dy> == <w, wgrad(x, dy)> == <x, dgrad(dy, w)> up to fp32 rounding."""
    Cin, Cout, k, stride, dil, hw = case
    gen = torch.Generator(device=DEV).manual_seed(sum(case[:5]))
    B, pad = 16, dil * (k - 1) // 2
    segs = Segs.make(B, hw)
    so = ops.conv_out_segs(segs, k, stride, pad, dil)
    x = torch.randn(segs.rows, Cin, device=DEV, generator=gen)
    w = torch.randn(Cout, Cin, k, k, device=DEV, generator=gen) / np.sqrt(Cin * k * k)
    dy = torch.randn(so.rows, Cout, device=DEV, generator=gen)
    y = ops.new_rows(so.rows, Cout, DEV)
    ops.conv_call(ops.Rows(x), segs, ops.pack_conv_weight_hip(w), y, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil)()
    s_y = float((y.buf.double() * dy.double()).sum())
    dw = ops.conv_wgrad(ops.Rows(x), ops.Rows(dy), segs, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, oihw=True)
    s_w = float((dw.double() * w.double()).sum())
    scale = float(y.buf.double().norm() * dy.double().norm())
    assert abs(s_y - s_w) / scale < 2e-6, (s_y, s_w, scale)
    if stride == 1:
        dx = ops.new_rows(segs.rows, Cin, DEV)
        ops.conv_call(ops.Rows(dy), so, ops.pack_conv_weight_hip(w, dgrad=True), dx, Cin=Cout, Cout=Cin, k=k, stride=1,
                      pad=dil * (k - 1) - pad, dil=dil)()
        s_x = float((dx.buf.double() * x.double()).sum())
        assert abs(s_y - s_x) / scale < 2e-6, (s_y, s_x, scale)


def test_dwconv_backward_adjoint_identity_full_size():
    gen = torch.Generator(device=DEV).manual_seed(3)
    B, C = 16, 512
    segs = Segs.make(B, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)])
    x = torch.randn(segs.rows, C, device=DEV, generator=gen)
    w = torch.randn(C, 1, 3, 3, device=DEV, generator=gen) / 3
    dy = torch.randn(segs.rows, C, device=DEV, generator=gen)
    y, dx = ops.new_rows(segs.rows, C, DEV), ops.new_rows(segs.rows, C, DEV)
    ops.dwconv3x3(ops.Rows(x), ops.pack_dw_weight(w), y, segs)
    ops.dwconv3x3(ops.Rows(dy), ops.pack_dw_weight(w.flip(2, 3)), dx, segs)
    dw = ops.dwconv3x3_wgrad(ops.Rows(x), ops.Rows(dy), segs, torch_layout=True)
    s_y = float((y.buf.double() * dy.double()).sum())
    scale = float(y.buf.double().norm() * dy.double().norm())
    assert abs(s_y - float((dw.double() * w.double()).sum())) / scale < 2e-6
    assert abs(s_y - float((dx.buf.double() * x.double()).sum())) / scale < 2e-6


@pytest.mark.parametrize("stride,down", [(1, False), (1, True), (2, True)])
def test_fused_bottleneck_autograd(stride, down):
    """train_ops.bottleneck (one autograd node; masks and the identity add ride in the conv epilogues) vs the stock module
    chain on the CPU: output, input gradient and all four weight gradients."""
    from pytorch_object_detection_amd import train_ops
    gen = torch.Generator().manual_seed(10 * stride + int(down))
    Cin, P = (128, 32) if not down else (64, 32)
    C4 = 4 * P

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(Cin, P, 1, bias=False)
            self.conv2 = torch.nn.Conv2d(P, P, 3, stride, 1, bias=False)
            self.conv3 = torch.nn.Conv2d(P, C4, 1, bias=False)
            self.bn1, self.bn2, self.bn3 = _frozen_bn(P, gen), _frozen_bn(P, gen), _frozen_bn(C4, gen)
            self.downsample = torch.nn.Sequential(torch.nn.Conv2d(Cin, C4, 1, stride, bias=False), _frozen_bn(C4, gen)) if down else None

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = F.relu(self.bn1(self.conv1(x)))
            y = F.relu(self.bn2(self.conv2(y)))
            return F.relu(self.bn3(self.conv3(y)) + idt)

    blk = Blk()
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d):
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * (2.0 / (m.weight.shape[1] * m.weight.shape[2] ** 2)) ** 0.5)
    blk.eval()
    x = torch.randn(2, Cin, 10, 12, generator=gen)
    xr = x.clone().requires_grad_(True)
    ref = blk(xr)
    dy = torch.randn(ref.shape, generator=gen)
    ref.backward(dy)
    b2 = copy.deepcopy(blk).to(DEV)
    for p in b2.parameters():
        p.grad = None
    xd = x.clone().to(DEV).requires_grad_(True)
    out = train_ops.bottleneck(b2, xd)
    seen, stack = set(), [out.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is not None and fn not in seen:
            seen.add(fn)
            stack.extend(f for f, _ in fn.next_functions)
    assert any(type(f).__name__ == "_BottleneckRowsBackward" for f in seen)
    out.backward(dy.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5, rtol=1e-5)
    pairs = [(xd.grad, xr.grad)] + [(getattr(b2, n).weight.grad, getattr(blk, n).weight.grad) for n in ("conv1", "conv2", "conv3")]
    if down:
        pairs.append((b2.downsample[0].weight.grad, blk.downsample[0].weight.grad))
    for a, b in pairs:
        s = float(b.abs().max())
        np.testing.assert_allclose(a.cpu().numpy() / s, b.numpy() / s, atol=3e-5)


def test_g9_full_width_layers_on_hip(golden):
    """SURVEY §8c G9 on the HIP path: the reference's full-width modules (fixtures generated by the real reference,
    tests/golden/make_golden.py g9; weights / inputs regenerated from tests/golden/lcg.py) -- HisBlock(256) through the plan
    builder, HISFCOSHead(256, 80) over two levels, and the standalone layer forwards of SEBlock / DepthWiseConv2d /
    PointWiseConv (the reference exposes them as ordinary layers, modules.py:40-49,65-73,107-121)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import lcg
    from pytorch_object_detection_amd import engine
    from pytorch_object_detection_amd.model.modules.modules import DepthWiseConv2d, PointWiseConv, ScaleExp, SEBlock
    from pytorch_object_detection_amd.model.od.HISFcos import HisBlock, HISFCOSHead
    g = golden("g9_full_width")

    def check(name, t):
        a = t.detach().cpu().numpy().reshape(-1)
        assert tuple(g[name + "_shape"]) == tuple(t.shape), name
        np.testing.assert_allclose(a[::7], g[name + "_s7"], atol=1e-4, rtol=1e-4, err_msg=name)
        ref_sum, ref_abs = g[name + "_sum"]
        assert abs(a.astype(np.float64).sum() - ref_sum) <= 2e-5 * ref_abs + 1e-3, name

    blk = HisBlock(256, 4, 2).eval(); lcg.fill_state(blk, 91); blk.to(DEV)
    x = to_rows(lcg.tensor((1, 256, 20, 20), 9101))
    plan = engine.Plan(torch.device(DEV))
    segs = Segs.make(1, [(20, 20)])
    out = ops.new_rows(400, 256, DEV)
    engine._his_block(plan, "b", blk, x, segs, out)
    plan.run()
    check("hisblock", from_rows(out, 1, 20, 20))
    head = HISFCOSHead(256, 80, 0.01).eval(); lcg.fill_state(head, 92); head.to(DEV)
    cls, cnt, reg = head([lcg.tensor((1, 256, 20, 20), 9201).to(DEV), lcg.tensor((1, 256, 10, 10), 9202).to(DEV)])
    for i in range(2):
        check(f"head_cls{i}", cls[i]); check(f"head_cnt{i}", cnt[i]); check(f"head_reg{i}", reg[i])
    with torch.no_grad():
        se = SEBlock(128, 4).eval(); lcg.fill_state(se, 93); se.to(DEV)
        check("se", se(lcg.tensor((2, 128, 20, 20), 9301).to(DEV)))
        dw = DepthWiseConv2d(512, 3).eval(); lcg.fill_state(dw, 94); dw.to(DEV)
        check("dw", dw(lcg.tensor((1, 512, 20, 20), 9401).to(DEV)))
        pw = PointWiseConv(2048, 256).eval(); lcg.fill_state(pw, 95); pw.to(DEV)
        check("pw", pw(lcg.tensor((1, 2048, 20, 20), 9501).to(DEV)))
        # shapes outside the autograd kernels: 5x5 stride-2 depthwise forward, ScaleExp
        dw5 = DepthWiseConv2d(64, 5, 2).eval().to(DEV)
        xx = torch.randn(2, 64, 13, 17)
        np.testing.assert_allclose(dw5(xx.to(DEV)).cpu().numpy(), F.conv2d(xx, dw5.weight.cpu(), None, 2, 2, 1, 64).numpy(), atol=1e-5, rtol=1e-5)
        sx = ScaleExp(1.2).to(DEV)
        np.testing.assert_allclose(sx(xx.to(DEV)).cpu().numpy(), torch.exp(xx * 1.2).numpy(), rtol=2e-6)
    with pytest.raises(Exception, match="not covered"):
        PointWiseConv(30, 8).to(DEV)(torch.randn(1, 30, 4, 4, device=DEV))
    # autograd through the standalone layers stays on the HIP nodes
    xg = torch.randn(2, 512, 8, 8, device=DEV).to(memory_format=torch.channels_last).requires_grad_(True)
    dw(xg).sum().backward()
    assert xg.grad is not None and dw.weight.grad is not None


@pytest.mark.parametrize("case", [(64, 128, 3, 2, 1, 13, 21), (128, 128, 3, 2, 1, 16, 16), (256, 128, 1, 2, 0, 10, 10), (64, 32, 1, 2, 0, 7, 9),
                                  (32, 64, 3, 2, 1, 2, 2)])
def test_conv_dgrad_strided_parity_classes(case):
    """Data gradient of the trunk's strided layers (3x3 s2 pad 1, 1x1 s2) on the conv kernel: one stride-1 launch per parity
    class of dX, outputs interleaved by the epilogue's scatter -- against torch autograd; with a folded BN scale and a ReLU
    mask in the epilogue as the fused bottleneck backward uses them."""
    Cin, Cout, k, s, pad, H, W = case
    gen = torch.Generator().manual_seed(sum(case))
    B = 2
    x = torch.randn(B, Cin, H, W, generator=gen).requires_grad_(True)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=gen) + 0.5
    y = F.conv2d(x, w * scale.view(-1, 1, 1, 1), None, s, pad)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    dx = torch.zeros(B * H * W, Cin, device=DEV)
    assert ops.conv_dgrad_strided(to_rows(dy), w.to(DEV), scale.to(DEV), ops.Rows(dx), B, H, W, k, s, pad)
    ref = x.grad
    sc = float(ref.abs().max())
    np.testing.assert_allclose(from_rows(ops.Rows(dx), B, H, W).numpy() / sc, ref.numpy() / sc, atol=2e-5)
    if k == 3:      # ReLU mask of the layer input applied in the class launches' epilogues
        mask_src = torch.randn(B, Cin, H, W, generator=gen)
        dx2 = torch.empty(B * H * W, Cin, device=DEV)
        assert ops.conv_dgrad_strided(to_rows(dy), w.to(DEV), scale.to(DEV), ops.Rows(dx2), B, H, W, k, s, pad, res=to_rows(mask_src), res_mask=True)
        np.testing.assert_allclose(from_rows(ops.Rows(dx2), B, H, W).numpy() / sc, (ref * (mask_src > 0)).numpy() / sc, atol=2e-5)


PATCH_CASES = [
    # Cin, Cout, dil, level sizes, act, res
    (256, 256, 1, [(20, 20)], ACT_RELU, False),
    (64, 128, 1, [(13, 21)], ACT_NONE, True),
    (256, 256, 2, [(20, 20)], ACT_SILU, False),            # HisBlock conv4 (dilation 2)
    (128, 512, 1, [(40, 40), (20, 20), (10, 10), (5, 5), (3, 3)], ACT_NONE, False),     # head tower over a pyramid
    (48, 80, 1, [(9, 95)], ACT_NONE, False),               # widest map the patch fits (W = 95), Cin % 32 != 0, Cout % 128 != 0
    (32, 5, 1, [(7, 7), (4, 4)], ACT_NONE, False),
    (256, 128, 2, [(1, 1), (2, 3)], ACT_RELU, True),
]


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("case", PATCH_CASES)
def test_conv3x3_patch_kernel(case, prec):
    """FD_TILE_128x128_PATCH: 3x3 stride-1 'same' conv with the (tile + halo) input patch staged once per channel chunk in
    LDS (north_star: 'LDS-staged input patches') against F.conv2d, on single maps and pyramids, through channel views."""
    from pytorch_object_detection_amd import _lib
    Cin, Cout, dil, hw, act, use_res = case
    gen = torch.Generator().manual_seed(Cin + Cout + dil + len(hw))
    B = 3
    xs = [torch.randn(B, Cin, h, w, generator=gen) for h, w in hw]
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    sc, sf = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen) * 0.1
    rs = [torch.randn(B, Cout, h, w, generator=gen) for h, w in hw]
    segs = Segs.make(B, hw)
    ref = []
    for x, r in zip(xs, rs):
        y = F.conv2d(x, wt, None, 1, dil, dil) * sc[None, :, None, None] + sf[None, :, None, None]
        if use_res:
            y = y + r
        ref.append(act_ref(y, act))
    xb = torch.full((segs.rows, Cin + 8), float("nan"), device=DEV)      # a channel slice of a wider buffer
    xb[:, 4:4 + Cin] = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(DEV)
    rb = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in rs]).contiguous().to(DEV)
    y = ops.new_rows(segs.rows, Cout, DEV)
    wp = (ops.pack_conv_weight_f16x3 if prec == "f16x3" else ops.pack_conv_weight)(wt.to(DEV))
    ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, scale=sc.to(DEV), shift=sf.to(DEV),
                  res=ops.Rows(rb) if use_res else None, act=act, tile=_lib.PATCH_TILE, precision=1 if prec == "f16x3" else 0)()
    got = y.tensor().cpu()
    for i, ((h, w), r) in enumerate(zip(hw, ref)):
        g = got[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, h, w, Cout).permute(0, 3, 1, 2)
        np.testing.assert_allclose(g.numpy(), r.numpy(), atol=ATOL, rtol=1e-4, err_msg=f"level {i}")
    # not a 3x3 'same' conv / too wide a map: a clean error, no launch
    with pytest.raises(Exception, match="PATCH|multi-level"):
        ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=0, dil=1, tile=_lib.PATCH_TILE)()


WINO_CASES = [
    # Cin, Cout, dil, level sizes, act, res
    (256, 256, 1, [(20, 20)], ACT_RELU, False),
    (64, 128, 1, [(13, 21)], ACT_NONE, True),                # odd sizes: half tiles at the right / bottom edge
    (256, 256, 2, [(20, 20)], ACT_SILU, False),              # HisBlock conv4 (dilation 2): four parity classes
    (128, 512, 1, [(40, 40), (20, 20), (10, 10), (5, 5), (3, 3)], ACT_NONE, False),     # head tower over a pyramid
    (48, 80, 1, [(9, 95)], ACT_NONE, False),                 # Cin % 32 != 0, Cout % 64 != 0 (cls_logits width)
    (8, 4, 1, [(7, 7), (4, 4)], ACT_NONE, False),
    (256, 128, 2, [(1, 1), (2, 3), (5, 7)], ACT_RELU, True), # dilation 2 on maps smaller than a tile
    (512, 512, 1, [(6, 6)], ACT_RELU, False),                # layer4 conv2: K = 512
    (64, 256, 1, [(80, 80)], ACT_RELU, True),                # wide and large enough for the 8-wave / 128-channel workgroups
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv3x3_winograd(case):
    """FD_TILE_WINOGRAD (fd_conv_wino.hip): F(2x2, 3x3) on the fp32 MFMA against F.conv2d -- pyramids, dilation 2 (parity classes),
    ragged sizes, channel views whose neighbours are NaN, BN fold + residual + activation epilogue.  Same bar as the direct kernel."""
    from pytorch_object_detection_amd import _lib
    Cin, Cout, dil, hw, act, use_res = case
    gen = torch.Generator().manual_seed(Cin + Cout + dil + len(hw))
    B = 3
    xs = [torch.randn(B, Cin, h, w, generator=gen) for h, w in hw]
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    sc, sf = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen) * 0.1
    rs = [torch.randn(B, Cout, h, w, generator=gen) for h, w in hw]
    segs = Segs.make(B, hw)
    ref = []
    for x, r in zip(xs, rs):
        y = F.conv2d(x, wt, None, 1, dil, dil) * sc[None, :, None, None] + sf[None, :, None, None]
        if use_res:
            y = y + r
        ref.append(act_ref(y, act))
    xb = torch.full((segs.rows, Cin + 8), float("nan"), device=DEV)      # a channel slice of a wider buffer
    xb[:, 4:4 + Cin] = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(DEV)
    rb = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in rs]).contiguous().to(DEV)
    yb = torch.full((segs.rows, Cout + 8), float("nan"), device=DEV)
    y = ops.Rows(yb, 4, Cout)
    wp = ops.pack_conv_weight_wino(wt.to(DEV))
    ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, scale=sc.to(DEV), shift=sf.to(DEV),
                  res=ops.Rows(rb) if use_res else None, act=act, tile=_lib.WINO_TILE)()
    got = yb.cpu()
    assert torch.isnan(got[:, :4]).all() and torch.isnan(got[:, 4 + Cout:]).all(), "wrote outside its channel view"
    got = got[:, 4:4 + Cout]
    for i, ((h, w), r) in enumerate(zip(hw, ref)):
        g = got[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, h, w, Cout).permute(0, 3, 1, 2)
        np.testing.assert_allclose(g.numpy(), r.numpy(), atol=ATOL, rtol=1e-4, err_msg=f"level {i}")
    if Cin >= 64:      # split-K: chunk loop divided over 2 / 4 workgroups per tile + the ordered combine launch (epilogue applied there)
        for ks in (2, 4):
            if Cin // 8 < 2 * ks:
                continue
            yb2 = torch.full((segs.rows, Cout + 8), float("nan"), device=DEV)
            ws = torch.empty(ks * segs.rows * ((Cout + 3) & ~3), device=DEV)
            run = ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb2, 4, Cout), Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, scale=sc.to(DEV),
                                shift=sf.to(DEV), res=ops.Rows(rb) if use_res else None, act=act, tile=_lib.WINO_TILE, ksplit=ks, workspace=ws)
            run()
            g2 = yb2.cpu()
            assert torch.isnan(g2[:, :4]).all() and torch.isnan(g2[:, 4 + Cout:]).all()
            np.testing.assert_allclose(g2[:, 4:4 + Cout].numpy(), got.numpy(), atol=2e-5, rtol=1e-5, err_msg=f"ksplit {ks}")
            first = g2.clone()
            run()
            assert torch.equal(yb2.cpu().nan_to_num(7.0), first.nan_to_num(7.0))          # bitwise reproducible
    with pytest.raises(Exception, match="WINOGRAD|multi-level"):     # not a stride-1 'same' 3x3: a clean error, no launch
        ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=0, dil=1, tile=_lib.WINO_TILE)()


def test_winograd_weight_pack_and_dgrad():
    """fd_wino_pack_weights_f32: U = G g G^T against a float64 einsum, in both modes; the mode-1 packing makes the kernel the
    data gradient of the conv (dX = conv(dY, flipped / transposed w * scale))."""
    from pytorch_object_detection_amd import _lib
    gen = torch.Generator().manual_seed(5)
    Cout, Cin = 40, 24
    w = torch.randn(Cout, Cin, 3, 3, generator=gen)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    U = torch.einsum("ir,ocrq,jq->ijoc", G, w.double(), G).reshape(16, Cout, Cin)
    got = ops.pack_conv_weight_wino(w.to(DEV)).cpu().reshape(2, Cin // 8, 16, 32, 8)      # [nb][cc][f][co][c]
    exp = torch.zeros(16, 64, Cin, dtype=torch.float64)
    exp[:, :Cout] = U
    exp[12:] = -exp[12:]          # the kernels form patch row 3 of B^T d B with the opposite sign: U carries the sign back
    exp = exp.reshape(16, 2, 32, Cin // 8, 8).permute(1, 3, 0, 2, 4)
    np.testing.assert_allclose(got.numpy(), exp.float().numpy(), rtol=0, atol=0)
    # data gradient through the same kernel
    B, H, W = 2, 9, 11
    scale = torch.rand(Cout, generator=gen) + 0.5
    x = torch.randn(B, Cin, H, W, generator=gen, requires_grad=True)
    dy = torch.randn(B, Cout, H, W, generator=gen)
    (F.conv2d(x, w, None, 1, 1, 1) * scale.view(1, -1, 1, 1)).backward(dy)
    segs = Segs.make(B, [(H, W)])
    dx = ops.new_rows(B * H * W, Cin, DEV)
    wp = ops.pack_conv_weight_wino(w.to(DEV), scale.to(DEV), dgrad=True)
    ops.conv_call(to_rows(dy), segs, wp, dx, Cin=Cout, Cout=Cin, k=3, pad=1, dil=1, tile=_lib.WINO_TILE)()
    np.testing.assert_allclose(from_rows(dx, B, H, W).numpy(), x.grad.numpy(), atol=ATOL, rtol=1e-4)


def test_winograd_vs_direct_kernel_at_the_bench_shapes():
    """BASELINE configs[1] sizes (16 x 640 x 640): the head tower (256 -> 512 over the five-level pyramid, 8-wave workgroups) and the
    dilation-2 HisBlock conv4 at 80 x 80 on the Winograd kernel against the direct kernel (bit-for-bit an fp32 fma chain, itself
    oracle-tested above) -- the CPU oracle would take minutes at this size; size-independent check: same result from two algorithms,
    bitwise identical from two launches."""
    from pytorch_object_detection_amd import _lib
    gen = torch.Generator().manual_seed(11)
    for name, hw, Cin, Cout, dil in (("tower", [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)], 256, 512, 1), ("HisBlock3.conv4", [(80, 80)], 256, 256, 2)):
        segs = Segs.make(16, hw)
        x = ops.Rows(torch.randn(segs.rows, Cin, generator=gen).to(DEV))
        w = (torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5).to(DEV)
        sc, sf = (torch.rand(Cout, generator=gen) + 0.5).to(DEV), (torch.randn(Cout, generator=gen) * 0.1).to(DEV)
        yd, yw, yw2 = (ops.new_rows(segs.rows, Cout, DEV) for _ in range(3))
        kw = dict(Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, scale=sc, shift=sf, act=ACT_RELU)
        ops.conv_call(x, segs, ops.pack_conv_weight(w), yd, tile=7, **kw)()
        wp = ops.pack_conv_weight_wino(w)
        ops.conv_call(x, segs, wp, yw, tile=_lib.WINO_TILE, **kw)()
        ops.conv_call(x, segs, wp, yw2, tile=_lib.WINO_TILE, **kw)()
        assert torch.equal(yw.tensor(), yw2.tensor()), name
        d = (yw.tensor() - yd.tensor()).abs().max().item()
        assert d < 5e-5, (name, d)            # (measured 1.3e-5 at unit-scale outputs; the parity bar vs the CPU path is 1e-4)
        assert float(yd.tensor().abs().max()) > 1.0


@pytest.mark.parametrize("case", [(144, 24, 9, 7), (192, 32, 16, 16), (816, 136, 5, 6), (288, 48, 12, 20), (96, 160, 8, 8)])
def test_conv1x1_with_input_gate(case):
    """fd_conv_params.gate: y = conv1x1(x * gate[n][c]) * scale + shift (+ res) -- the MBConv project conv with the squeeze-excitation gate
    folded into its loader (EfficientNet widths incl. Cin % 32 != 0), against the two-step torch reference; and the gates-only mode of
    fd_se_scale_nhwc that produces them."""
    Cin, Cout, H, W = case
    gen = torch.Generator().manual_seed(Cin + Cout)
    B = 3
    x = torch.randn(B, Cin, H, W, generator=gen)
    gate = torch.rand(B, Cin, generator=gen)
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / Cin ** 0.5
    scale, shift = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, W, generator=gen)
    ref = F.conv2d(x * gate[:, :, None, None], w) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res
    segs = Segs.make(B, [(H, W)])
    gbuf = torch.full((B, Cin + 4), float("nan"), device=DEV)           # gate rows with a stride (gate_cs > Cin)
    gbuf[:, :Cin] = gate.to(DEV)
    y = ops.new_rows(B * H * W, Cout, DEV)
    ops.conv_call(to_rows(x), segs, ops.pack_conv_weight(w.to(DEV)), y, Cin=Cin, Cout=Cout, k=1, scale=scale.to(DEV), shift=shift.to(DEV),
                  res=to_rows(res), gate=gbuf[:, :Cin])()
    np.testing.assert_allclose(from_rows(y, B, H, W).numpy(), ref.numpy(), atol=ATOL, rtol=1e-4)
    with pytest.raises(Exception, match="gate"):            # 3x3: not a GEMM layer
        w3 = torch.randn(Cout, Cin, 3, 3, generator=gen)
        ops.conv_call(to_rows(x), segs, ops.pack_conv_weight(w3.to(DEV)), y, Cin=Cin, Cout=Cout, k=3, pad=1, gate=gbuf[:, :Cin])()
    # gates-only squeeze-excitation
    Cr = max(4, Cin // 24)
    w1, b1 = torch.randn(Cr, Cin, generator=gen) / Cin ** 0.5, torch.randn(Cr, generator=gen) * 0.1
    w2, b2 = torch.randn(Cin, Cr, generator=gen) / Cr ** 0.5, torch.randn(Cin, generator=gen) * 0.1
    m = x.mean(dim=(2, 3))
    g_ref = torch.sigmoid(F.silu(m @ w1.t() + b1) @ w2.t() + b2)
    ws = ops.se_workspace(B, H * W, Cin, DEV)
    xr = to_rows(x)
    before = xr.tensor().clone()
    g = ops.se_gate(xr, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), B, H * W, Cr, ws)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref.numpy(), atol=1e-5, rtol=1e-5)
    assert torch.equal(xr.tensor(), before)                 # x untouched


@pytest.mark.parametrize("k,stride,pad", [(1, 1, 0), (3, 1, 1), (3, 2, 1)])
def test_autotune_force_mode_times_only_tiles_the_layer_has(k, stride, pad, monkeypatch):
    """ops.autotune_conv's live-timing path (FD_AUTOTUNE=1 / force) used to offer FD_TILE_128x128_PATCH -- a 3x3 stride-1 'same'
    only kernel -- to every layer and died on the library's FD_E_UNSUPPORTED.  Force mode on a 1x1, a 3x3 and a strided 3x3 layer: a tile
    is chosen, the launch under it is correct, and no table entry is written to disk."""
    monkeypatch.setattr(ops, "_TUNE_MODE", "force")
    monkeypatch.setattr(ops, "_TUNE_CACHE", {})
    monkeypatch.setattr(ops, "_TUNE_LOADED", True)
    gen = torch.Generator().manual_seed(40 + k + stride)
    B, Cin, Cout, H, W = 2, 64, 128, 24, 20
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    ref = F.relu(F.conv2d(x, w, None, stride, pad))
    segs = Segs.make(B, [(H, W)])
    so = ops.conv_out_segs(segs, k, stride, pad, 1)
    y = ops.new_rows(so.rows, Cout, DEV)
    ws = torch.empty(ops.KSPLIT_MAX * so.rows * Cout, device=DEV)
    call = ops.conv_call(to_rows(x), segs, ops.pack_conv_weight(w.to(DEV)), y, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, act=ACT_RELU,
                         workspace=ws)
    code = ops.autotune_conv(call, f"test|k{k}s{stride}", so.rows, Cout, (Cin // 32) * k * k, reps=2)
    assert (code & 0xFF) in _TILE_IDS or (code & 0xFF) == 0
    assert (code & 0xFF) != 13 or (k == 3 and stride == 1)
    y.buf.fill_(float("nan"))
    call()
    Ho, Wo = so.level_hw()[0]
    np.testing.assert_allclose(from_rows(y, B, Ho, Wo).numpy(), ref.numpy(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("case", [
    # Cin, Cout, H, W, residual, act, act_c0, res_mask
    (64, 256, 20, 24, True, ACT_RELU, 0, False),        # bottleneck conv3 (+ residual + ReLU)
    (256, 64, 20, 24, False, ACT_RELU, 0, False),       # bottleneck conv1
    (2048, 256, 5, 7, False, ACT_RELU, 0, False),       # FPN lateral, ragged M tail (70 rows)
    (256, 256, 9, 11, False, ACT_SILU, 128, False),     # HisBlock conv1+2: SiLU on the upper half only
    (512, 96, 6, 6, True, ACT_NONE, 0, True),           # Cout % 64 == 32, ReLU-mask residual (a data gradient)
    (32, 64, 3, 3, False, ACT_NONE, 0, False),          # one K-tile
])
def test_conv1x1_wave_tile_is_bit_identical_to_the_workgroup_kernel(case):
    """FD_TILE_WAVE64 (fd_conv_wave.hip: one wave = one 64 x 64 tile, no barrier in the K loop, weights in MFMA fragment order): the same
    fma chain per output as the workgroup-tiled kernel -> bitwise equal results, on channel-slice views whose neighbours are NaN."""
    from pytorch_object_detection_amd import _lib
    Cin, Cout, H, W, use_res, act, act_c0, res_mask = case
    gen = torch.Generator().manual_seed(sum(case[:4]))
    B = 3
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)
    scale, shift = (torch.rand(Cout, generator=gen) + 0.5).to(DEV), torch.randn(Cout, generator=gen).to(DEV)
    segs = Segs.make(B, [(H, W)])
    rows = B * H * W
    xb = torch.full((rows, Cin + 8), float("nan"), device=DEV)
    xb[:, 4:4 + Cin] = x.permute(0, 2, 3, 1).reshape(rows, Cin).to(DEV)
    xr = ops.Rows(xb, 4, Cin)
    res = ops.Rows(torch.randn(rows, Cout + 4, generator=gen).to(DEV), 4, Cout) if use_res else None
    wd = w.to(DEV)
    outs = []
    for tile in (4, _lib.WAVE_TILE):
        yb = torch.full((rows, Cout + 8), float("nan"), device=DEV)
        y = ops.Rows(yb, 4, Cout)
        ops.conv_call(xr, segs, ops.pack_conv_weight(wd), y, Cin=Cin, Cout=Cout, k=1, scale=scale, shift=shift, res=res, act=act, act_c0=act_c0,
                      res_mask=res_mask, tile=tile, w_frag=ops.pack_conv_weight_wave(wd))()
        assert torch.isnan(yb[:, :4]).all() and torch.isnan(yb[:, 4 + Cout:]).all()       # nothing outside the channel view was written
        outs.append(y.tensor().clone())
    assert torch.equal(outs[0], outs[1])
    ref = F.conv2d(x, w) * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1)
    if use_res:
        r = res.tensor().cpu().reshape(B, H, W, Cout).permute(0, 3, 1, 2)
        ref = torch.where(r > 0, ref, torch.zeros_like(ref)) if res_mask else ref + r
    if act != ACT_NONE:
        a = act_ref(ref, act)
        ref = torch.cat([ref[:, :act_c0], a[:, act_c0:]], 1)
    got = outs[1].reshape(B, H, W, Cout).permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=ATOL, rtol=RTOL)


# ------------------------------------------------------------------------------------------------ GroupNorm fused into its neighbours
@pytest.mark.parametrize("kind", ["tile4", "tile8", "tile9", "wave", "winograd", "winograd_d2"])
@pytest.mark.parametrize("G", [32, 64])
def test_conv_epilogue_row_group_statistics(kind, G):
    """fd_conv_params.gn_stats: per output row and channel group the (sum, sum of squares) of the stored values, from the epilogue of every
    kernel that offers it -- against the same sums taken from the conv's own output, on a ragged pyramid (rows past M, tiles straddling
    images and levels); the conv output itself is bitwise what it is without the statistics."""
    from pytorch_object_detection_amd import _lib
    gen = torch.Generator().manual_seed(G + len(kind))
    B, Cin, Cout = 3, 64, 256
    hw = [(9, 11), (5, 6), (3, 3), (1, 2)]
    wino = kind.startswith("winograd")
    dil = 2 if kind.endswith("d2") else 1
    k = 3 if wino else 1
    segs = Segs.make(B, hw)
    x = ops.Rows(torch.randn(segs.rows, Cin, generator=gen).to(DEV))
    w = (torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)).to(DEV)
    res = ops.Rows(torch.randn(segs.rows, Cout, generator=gen).to(DEV))
    shift = torch.randn(Cout, generator=gen).to(DEV)
    tile = {"tile4": 4, "tile8": 8, "tile9": 9, "wave": _lib.WAVE_TILE}.get(kind, _lib.WINO_TILE)
    wp = ops.pack_conv_weight_wino(w) if wino else ops.pack_conv_weight(w)
    wf = ops.pack_conv_weight_wave(w) if kind == "wave" else None
    outs = []
    for with_stats in (False, True):
        y = ops.new_rows(segs.rows, Cout, DEV)
        rgs = torch.full((segs.rows, G, 2), float("nan"), device=DEV) if with_stats else None
        ops.conv_call(x, segs, wp, y, Cin=Cin, Cout=Cout, k=k, pad=dil if wino else 0, dil=dil, shift=shift, res=res, act=ACT_RELU, tile=tile,
                      w_frag=wf, gn_stats=rgs, gn_groups=G)()
        outs.append(y.tensor().clone())
    assert torch.equal(outs[0], outs[1])
    yv = outs[1].double().view(segs.rows, G, Cout // G)
    np.testing.assert_allclose(rgs[..., 0].cpu().numpy(), yv.sum(-1).cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rgs[..., 1].cpu().numpy(), (yv * yv).sum(-1).cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_groupnorm_from_rowstats_matches_the_three_pass_groupnorm():
    """fd_groupnorm_from_rowstats + fd_groupnorm_apply_nhwc on exact row-group sums == fd_groupnorm_act_nhwc (statistics / finalise / normalise
    over the map itself) to rounding, per (level, image); the coefficient table is the affine the apply kernel uses."""
    gen = torch.Generator().manual_seed(77)
    B, Cc, G = 3, 512, 32
    hw = [(12, 9), (5, 7), (2, 3), (1, 1)]
    segs = Segs.make(B, hw)
    x = torch.randn(segs.rows, Cc, generator=gen) * 2 + 0.5
    gamma, beta = (torch.rand(Cc, generator=gen) + 0.5).to(DEV), torch.randn(Cc, generator=gen).to(DEV)
    xd = ops.Rows(x.to(DEV))
    ws0 = ops.groupnorm_workspace(segs, G, DEV)
    y0 = ops.new_rows(segs.rows, Cc, DEV)
    ops.groupnorm_act(xd, gamma, beta, y0, segs, G, ACT_SILU, ws0)
    xv = x.view(segs.rows, G, Cc // G)
    rgs = torch.stack([xv.sum(-1), (xv * xv).sum(-1)], -1).contiguous().to(DEV)
    ws1 = ops.groupnorm_workspace(segs, G, DEV)
    coef = torch.empty(segs.nseg * B, 2, Cc, device=DEV)
    ops.groupnorm_from_rowstats(rgs, Cc, G, 1e-5, gamma, beta, segs, ws1, coef)
    y1 = ops.new_rows(segs.rows, Cc, DEV)
    ops.groupnorm_apply(xd, gamma, beta, y1, segs, G, ACT_SILU, ws1)
    np.testing.assert_allclose(y1.tensor().cpu().numpy(), y0.tensor().cpu().numpy(), atol=2e-5, rtol=2e-5)
    # coef: y = silu(x * a + b) row by row
    img = 0
    for lv, (h, w) in enumerate(hw):
        for n in range(B):
            r0 = segs.m_start[lv] + n * h * w
            a, b = coef[lv * B + n, 0].cpu(), coef[lv * B + n, 1].cpu()
            ref = F.silu(x[r0:r0 + h * w] * a + b)
            np.testing.assert_allclose(y1.tensor()[r0:r0 + h * w].cpu().numpy(), ref.numpy(), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("Cc", [512, 128])
def test_dwconv3x3_with_fused_groupnorm_input_and_output_statistics(Cc):
    """fd_dwconv3x3_gn_nhwc: depthwise 3x3 over act(x * a + b) with the zero padding applied AFTER the affine (the reference pads the
    normalised map), plus the row-group sums of its output -- against torch on a ragged pyramid."""
    gen = torch.Generator().manual_seed(Cc)
    B, G = 2, 32
    hw = [(10, 13), (4, 5), (2, 2), (1, 1)]
    segs = Segs.make(B, hw)
    xs = [torch.randn(B, Cc, h, w, generator=gen) for h, w in hw]
    wt = torch.randn(Cc, 1, 3, 3, generator=gen) / 3
    coef = torch.stack([torch.rand(len(hw) * B, Cc, generator=gen) + 0.5, torch.randn(len(hw) * B, Cc, generator=gen)], 1).contiguous()
    xr = ops.Rows(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cc) for t in xs]).to(DEV))
    y = ops.new_rows(segs.rows, Cc, DEV)
    rgs = torch.full((segs.rows, G, 2), float("nan"), device=DEV)
    ops.dwconv3x3_gn(xr, ops.pack_dw_weight(wt.to(DEV)), y, segs, coef.to(DEV), ACT_RELU, rgs, G)
    got = y.tensor().cpu()
    for lv, ((h, w), x) in enumerate(zip(hw, xs)):
        a = coef[lv * B:(lv + 1) * B, 0].view(B, Cc, 1, 1)
        b = coef[lv * B:(lv + 1) * B, 1].view(B, Cc, 1, 1)
        ref = F.conv2d(F.relu(x * a + b), wt, None, 1, 1, 1, Cc)
        g = got[segs.m_start[lv]:segs.m_start[lv + 1]].reshape(B, h, w, Cc).permute(0, 3, 1, 2)
        np.testing.assert_allclose(g.numpy(), ref.numpy(), atol=1e-5, rtol=1e-5, err_msg=f"level {lv}")
    gv = got.double().view(segs.rows, G, Cc // G)
    np.testing.assert_allclose(rgs[..., 0].cpu().numpy(), gv.sum(-1).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rgs[..., 1].cpu().numpy(), (gv * gv).sum(-1).numpy(), rtol=1e-5, atol=1e-5)


def test_conv1x1_with_groupnorm_affine_and_silu_in_its_loader():
    """fd_conv_params.gate + gate_b + gate_act over a PYRAMID: y = conv1x1(silu(x * a[img] + b[img])) + bias + residual, the form HISFCOSHead's
    pw2 takes when GroupNorm 2 is folded into it (HISFcos.py:220-222)."""
    gen = torch.Generator().manual_seed(5)
    B, Cin, Cout = 2, 512, 256
    hw = [(9, 12), (4, 6), (2, 3), (1, 1)]
    segs = Segs.make(B, hw)
    xs = [torch.randn(B, Cin, h, w, generator=gen) for h, w in hw]
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)
    bias = torch.randn(Cout, generator=gen)
    coef = torch.stack([torch.rand(len(hw) * B, Cin, generator=gen) + 0.5, torch.randn(len(hw) * B, Cin, generator=gen)], 1).contiguous().to(DEV)
    rs = [torch.randn(B, Cout, h, w_, generator=gen) for h, w_ in hw]
    xr = ops.Rows(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(DEV))
    rr = ops.Rows(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in rs]).to(DEV))
    y = ops.new_rows(segs.rows, Cout, DEV)
    ops.conv_call(xr, segs, ops.pack_conv_weight(w.to(DEV)), y, Cin=Cin, Cout=Cout, k=1, shift=bias.to(DEV), res=rr, gate=coef[:, 0], gate_b=coef[:, 1],
                  gate_act=ACT_SILU)()
    got = y.tensor().cpu()
    cc = coef.cpu()
    for lv, ((h, w_), x, r) in enumerate(zip(hw, xs, rs)):
        a = cc[lv * B:(lv + 1) * B, 0].view(B, Cin, 1, 1)
        b = cc[lv * B:(lv + 1) * B, 1].view(B, Cin, 1, 1)
        ref = F.conv2d(F.silu(x * a + b), w, bias) + r
        g = got[segs.m_start[lv]:segs.m_start[lv + 1]].reshape(B, h, w_, Cout).permute(0, 3, 1, 2)
        np.testing.assert_allclose(g.numpy(), ref.numpy(), atol=ATOL, rtol=1e-4, err_msg=f"level {lv}")


def test_hisfcos_head_fused_groupnorm_equals_unfused_and_oracle(monkeypatch):
    """HISFCOSHead(256, 80) on a five-level pyramid: the plan with GroupNorm folded into pw1 / dw1 / pw2 / the tower (engine.GN_FUSED) against
    the three-pass GroupNorm plan and against the oracle's restatement of HISFcos.py:211-229."""
    from oracle import torch_ref as R
    from pytorch_object_detection_amd import engine
    from pytorch_object_detection_amd.model.od.HISFcos import HISFCOSHead
    torch.manual_seed(12)
    head = HISFCOSHead(256, 80).eval()
    with torch.no_grad():
        for n, p_ in head.named_parameters():
            if p_.dim() == 4:
                p_.copy_(torch.randn(p_.shape) * (2.0 / (p_.shape[1] * p_.shape[2] * p_.shape[3])) ** 0.5)
            elif "gn" in n or ".1." in n:
                p_.copy_(torch.rand(p_.shape) + 0.5 if n.endswith("weight") else torch.randn(p_.shape) * 0.3)
    sd = {"head." + k: v.clone() for k, v in head.state_dict().items()}
    feats = [torch.randn(2, 256, s, s + 1) for s in (20, 10, 5, 3, 1)]
    with torch.no_grad():
        ref = R.his_head(sd, feats)
    head.to(DEV)
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(engine, "GN_FUSED", fused)
        head.invalidate_plans()
        o = head([f.to(DEV) for f in feats])
        outs[fused] = [t.clone() for grp in o for t in grp]
        names = head._plans[next(iter(head._plans))][1][0].names
        assert ("head.gn1" in names) == (not fused) and ("head.gn1.stats" in names) == fused
    for a, b in zip(outs[True], outs[False]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=3e-5, rtol=3e-5)
    for a, r in zip(outs[True], [t for grp in ref for t in grp]):
        np.testing.assert_allclose(a.cpu().numpy(), r.numpy(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("case", WINO_CASES + [(256, 80, 1, [(17, 23), (9, 12), (5, 6), (3, 3), (2, 1)], ACT_EXP, False),
                                            (64, 64, 2, [(21, 18), (7, 9)], ACT_NONE, False)])      # dilation 2, odd sizes: ragged parity classes
def test_conv3x3_winograd_f4x4(case):
    """FD_TILE_WINOGRAD4 (fd_conv_wino4.hip): F(4x4, 3x3) on the fp32 MFMA against F.conv2d -- pyramids, ragged sizes (partial 4x4 tiles at the
    right / bottom edges, maps smaller than a tile), dilation 2 (four parity classes per image), channel views whose neighbours are NaN, BN fold +
    residual + activation epilogue, per-level ScaleExp."""
    from pytorch_object_detection_amd import _lib
    Cin, Cout, dil, hw, act, use_res = case
    gen = torch.Generator().manual_seed(2 * Cin + Cout + len(hw))
    B = 3
    xs = [torch.randn(B, Cin, h, w, generator=gen) for h, w in hw]
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    sc, sf = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen) * 0.1
    rs = [torch.randn(B, Cout, h, w, generator=gen) for h, w in hw]
    prm = [1.2, 0.9, 1.1, 0.8, 1.0][:len(hw)]
    segs = Segs.make(B, hw)
    ref = []
    for lv, (x, r) in enumerate(zip(xs, rs)):
        y = F.conv2d(x.double(), wt.double(), None, 1, dil, dil) * sc.double()[None, :, None, None] + sf.double()[None, :, None, None]
        if use_res:
            y = y + r.double()
        if act == ACT_EXP:
            y = torch.cat([y[:, :4], torch.exp(y[:, 4:] * prm[lv])], 1)
        elif act != ACT_NONE:
            y = act_ref(y, act)
        ref.append(y.float())
    xb = torch.full((segs.rows, Cin + 8), float("nan"), device=DEV)
    xb[:, 4:4 + Cin] = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(DEV)
    rb = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in rs]).contiguous().to(DEV)
    yb = torch.full((segs.rows, Cout + 8), float("nan"), device=DEV)
    y = ops.Rows(yb, 4, Cout)
    wp = ops.pack_conv_weight_wino4(wt.to(DEV))
    run = ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, scale=sc.to(DEV), shift=sf.to(DEV),
                        res=ops.Rows(rb) if use_res else None, act=act, act_c0=4 if act == ACT_EXP else 0, seg_param=prm if act == ACT_EXP else None,
                        tile=_lib.WINO4_TILE)
    run()
    got = yb.cpu()
    assert torch.isnan(got[:, :4]).all() and torch.isnan(got[:, 4 + Cout:]).all(), "wrote outside its channel view"
    got = got[:, 4:4 + Cout]
    assert not torch.isnan(got).any(), "an output pixel was never written"
    for i, ((h, w), r) in enumerate(zip(hw, ref)):
        g = got[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, h, w, Cout).permute(0, 3, 1, 2)
        # F(4x4)'s transforms multiply by up to 8: the fp32 evaluation of ONE layer is good to ~2e-5 of the output scale (max over 10^6 outputs:
        # 8e-5 at |y| <= 5 in the CPU emulation tools/wino44_emul.py; mean 2e-6), a whole model to 2e-5 * (1 + |y|)
        scale = float(r.abs().max()) + 1.0
        err = (g - r).abs()
        assert float(err.max()) < 4e-5 * scale and float(err.mean()) < 2e-6 * scale, (i, float(err.max()), float(err.mean()), scale)
    first = yb.clone()
    run()
    assert torch.equal(yb.nan_to_num(7.0), first.nan_to_num(7.0))          # bitwise reproducible
    if Cin >= 32:        # split-K (maps with few tiles): raw partial outputs per slice -> workspace -> the combine launch applies the epilogue
        ws = torch.empty(2 * segs.rows * ((Cout + 3) & ~3), dtype=torch.float32, device=DEV)
        yb2 = torch.full((segs.rows, Cout + 8), float("nan"), device=DEV)
        ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb2, 4, Cout), Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, scale=sc.to(DEV), shift=sf.to(DEV),
                      res=ops.Rows(rb) if use_res else None, act=act, act_c0=4 if act == ACT_EXP else 0, seg_param=prm if act == ACT_EXP else None,
                      tile=_lib.WINO4_TILE, ksplit=2, workspace=ws)()
        a2, a1 = yb2[:, 4:4 + Cout].cpu(), first[:, 4:4 + Cout].cpu()
        assert torch.isnan(yb2[:, :4]).all() and not torch.isnan(a2).any()
        tol = 2e-5 * (float(a1.abs().max()) + 1.0)
        assert float((a2 - a1).abs().max()) < tol, float((a2 - a1).abs().max())       # same products, another summation order
    with pytest.raises(Exception, match="WINOGRAD4"):     # dilation 3: a clean error, no launch
        ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=3, dil=3, tile=_lib.WINO4_TILE)()


def test_winograd_f4x4_data_gradient_packing():
    """ops.pack_conv_weight_wino4(dgrad=True): the data gradient of a 3x3 stride-1 conv as a FD_TILE_WINOGRAD4 launch on dY with the flipped /
    transposed weights times the per-Cout scale of a folded BatchNorm (what train_ops.PACKS hands _conv_launch), against conv_transpose2d."""
    from pytorch_object_detection_amd import _lib
    gen = torch.Generator().manual_seed(77)
    B, Cin, Cout, hw = 2, 64, 96, [(19, 14), (6, 5)]
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    sc = torch.rand(Cout, generator=gen) + 0.5
    dys = [torch.randn(B, Cout, h, w, generator=gen) for h, w in hw]
    segs = Segs.make(B, hw)
    dyb = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in dys]).contiguous().to(DEV)
    dxb = torch.full((segs.rows, Cin), float("nan"), device=DEV)
    wp = ops.pack_conv_weight_wino4(wt.to(DEV), sc.to(DEV), dgrad=True)
    ops.conv_call(ops.Rows(dyb), segs, wp, ops.Rows(dxb), Cin=Cout, Cout=Cin, k=3, pad=1, dil=1, tile=_lib.WINO4_TILE)()
    got = dxb.cpu()
    for i, ((h, w), dy) in enumerate(zip(hw, dys)):
        ref = F.conv_transpose2d(dy.double() * sc.double()[None, :, None, None], wt.double(), None, 1, 1).float()
        g = got[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, h, w, Cin).permute(0, 3, 1, 2)
        scale = float(ref.abs().max()) + 1.0
        err = (g - ref).abs()
        assert float(err.max()) < 4e-5 * scale and float(err.mean()) < 2e-6 * scale, (i, float(err.max()), float(err.mean()))


def test_winograd_f4x4_random_shapes_against_the_direct_kernel():
    """Seeded random layer shapes (ragged pyramids, maps smaller than a tile, dilation 1 / 2, Cin any multiple of 8, Cout any multiple of 4 incl. widths whose
    last 32-cout block is dead, split-K with an uneven last slice, residual add / ReLU mask, channel views inside wider buffers): FD_TILE_WINOGRAD4 against the
    direct implicit-GEMM kernel (an fp32 fma chain pinned against the oracle elsewhere) on identical inputs."""
    from pytorch_object_detection_amd import _lib
    import random
    rng = random.Random(20261004)
    gen = torch.Generator().manual_seed(99)
    for case in range(24):
        Cin = 8 * rng.randint(1, 20)
        Cout = 4 * rng.randint(1, 40)
        dil = rng.choice((1, 1, 2))
        nlev = rng.randint(1, 4)
        hw = [(rng.randint(1, 23), rng.randint(1, 23)) for _ in range(nlev)]
        B = rng.randint(1, 3)
        use_res, res_mask = rng.random() < 0.4, rng.random() < 0.3
        act = rng.choice((ACT_NONE, ACT_RELU, ACT_SILU))
        ks = rng.choice((1, 1, 2, 3)) if Cin >= 48 else 1
        xo, yo = 4 * rng.randint(0, 2), 4 * rng.randint(0, 2)
        segs = Segs.make(B, hw)
        xb = torch.full((segs.rows, Cin + xo + 4), float("nan"), device=DEV)
        xb[:, xo:xo + Cin] = torch.randn(segs.rows, Cin, generator=gen).to(DEV)
        wt = (torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5).to(DEV)
        sc, sf = (torch.rand(Cout, generator=gen) + 0.5).to(DEV), (torch.randn(Cout, generator=gen) * 0.1).to(DEV)
        rb = torch.randn(segs.rows, Cout, generator=gen).to(DEV)
        outs = []
        for tile, wp in ((_lib.WINO4_TILE, ops.pack_conv_weight_wino4(wt)), (0, ops.pack_conv_weight(wt))):
            yb = torch.full((segs.rows, Cout + yo + 4), float("nan"), device=DEV)
            ws = torch.empty(max(ks, 1) * segs.rows * ((Cout + 3) & ~3), dtype=torch.float32, device=DEV) if (ks > 1 and tile) else None
            ops.conv_call(ops.Rows(xb, xo, Cin), segs, wp, ops.Rows(yb, yo, Cout), Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, scale=sc, shift=sf,
                          res=ops.Rows(rb) if use_res else None, res_mask=use_res and res_mask, act=act, tile=tile,
                          ksplit=ks if (ks > 1 and tile and Cin // 8 >= 2 * ks) else 1, workspace=ws)()
            assert torch.isnan(yb[:, :yo]).all() and torch.isnan(yb[:, yo + Cout:]).all(), (case, "wrote outside its channel view")
            outs.append(yb[:, yo:yo + Cout].cpu())
        a, b = outs
        assert not torch.isnan(a).any(), (case, "an output was never written")
        scale = float(b.abs().max()) + 1.0
        if use_res and res_mask:          # (a ReLU-mask residual exactly at 0 decides by sign only: identical inputs, identical masks)
            pass
        err = (a - b).abs()
        assert float(err.max()) < 6e-5 * scale, (case, Cin, Cout, dil, hw, B, ks, float(err.max()), scale)


@pytest.mark.parametrize("case", [(256, 5, [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)], 2), (64, 5, [(17, 23), (9, 12), (5, 6), (3, 3), (2, 1)], 3),
                                  (32, 1, [(33, 16)], 2), (16, 2, [(16, 16), (1, 1)], 1), (48, 4, [(7, 40)], 2), (32, 8, [(18, 18), (15, 17)], 2), (32, 7, [(9, 9)], 1)])
def test_conv3x3_narrow_on_the_vector_unit(case):
    """FD_TILE_NARROW (fd_conv_narrow.hip): 3x3 stride-1 pad-1 convs of <= 8 output channels as exact fp32 FMA chains, one thread per output pixel,
    against F.conv2d in fp64 -- the bench's cnt_logits + reg_pred shape (pyramid of 640 x 640, 256 -> 5), ragged pyramids (partial 16 x 16 tiles, maps
    smaller than a tile), every padded width (1, 2, 4, 5, 8), channel views whose neighbours are NaN, bias + per-level ScaleExp on channels >= 1."""
    from pytorch_object_detection_amd import _lib
    Cin, Cout, hw, B = case
    gen = torch.Generator().manual_seed(3 * Cin + Cout + len(hw))
    xs = [torch.randn(B, Cin, h, w, generator=gen) for h, w in hw]
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=gen) * 0.3
    prm = [1.2, 0.9, 1.1, 0.8, 1.0][:len(hw)]
    segs = Segs.make(B, hw)
    ref = []
    for lv, x in enumerate(xs):
        y = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1)
        ref.append(torch.cat([y[:, :1], torch.exp(y[:, 1:] * prm[lv])], 1).float())
    xb = torch.full((segs.rows, Cin + 8), float("nan"), device=DEV)
    xb[:, 4:4 + Cin] = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(DEV)
    yb = torch.full((segs.rows, Cout + 3), float("nan"), device=DEV)
    y = ops.Rows(yb, 2, Cout)                       # (the output view needs no alignment: scalar stores)
    wp = ops.pack_conv_weight_narrow(wt.to(DEV))
    assert wp.shape == (Cin // 16, 3, 4, 3, 4, 8) and _lib.lib().fd_conv_narrow_nco(Cout) in (4, 5, 8)
    run = ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=1, shift=bias.to(DEV), act=ACT_EXP, act_c0=1, seg_param=prm,
                        tile=_lib.NARROW_TILE)
    run()
    got = yb.cpu()
    assert torch.isnan(got[:, :2]).all() and torch.isnan(got[:, 2 + Cout:]).all(), "wrote outside its channel view"
    got = got[:, 2:2 + Cout]
    assert not torch.isnan(got).any(), "an output pixel was never written"
    for i, ((h, w), r) in enumerate(zip(hw, ref)):
        g = got[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, h, w, Cout).permute(0, 3, 1, 2)
        np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=2e-5, atol=5e-6, err_msg=f"level {i}")      # one fp32 fma chain of up to 2 304 terms per output (+ expf)
    first = yb.clone()
    run()
    assert torch.equal(yb.nan_to_num(7.0), first.nan_to_num(7.0))          # bitwise reproducible
    with pytest.raises(Exception, match="NARROW"):       # 9 output channels / a residual: a clean error, no launch
        ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(torch.empty(segs.rows, 12, device=DEV), 0, 9), Cin=Cin, Cout=9, k=3, pad=1, tile=_lib.NARROW_TILE)()


@pytest.mark.parametrize("case", [(256, 5, 512, 64, [(40, 40), (20, 20), (10, 10), (5, 5), (3, 3)], 2, ACT_RELU), (32, 8, 64, 16, [(17, 23), (2, 1)], 3, ACT_SILU),
                                  (64, 4, 128, 8, [(33, 16)], 1, ACT_NONE)])
def test_groupnorm_split_between_a_slice_pass_and_the_narrow_loader(case):
    """The head tower's GroupNorm + ReLU (HISFcos.py:221-225) without a normalise pass over the box half: fd_groupnorm_stats_nhwc leaves the statistics
    of the WHOLE map and the per-(level, image, channel) affine; fd_coef_apply_nhwc normalises the class slice; the FD_TILE_NARROW predictor applies the
    affine + activation of ITS slice to the patch on its way to LDS (fd_conv_params.gate / gate_b), padding zero after the normalisation.  Both must
    equal, BIT FOR BIT, the three-pass fd_groupnorm_act_nhwc followed by the ungated predictor; the statistics themselves are checked against
    F.group_norm in fp64."""
    from pytorch_object_detection_amd import _lib
    Cin, Cout, Ct, G, hw, B, act = case                       # the predictor reads the LAST Cin channels of a Ct-channel map normalised in G groups
    gen = torch.Generator().manual_seed(Cin + Cout + Ct + len(hw))
    segs = Segs.make(B, hw)
    xs = [torch.randn(B, Ct, h, w, generator=gen) * 1.7 + 0.4 for h, w in hw]
    gamma, beta = torch.rand(Ct, generator=gen) + 0.5, torch.randn(Ct, generator=gen) * 0.3
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=gen) * 0.3
    raw = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Ct) for t in xs]).to(DEV).contiguous()
    wp = ops.pack_conv_weight_narrow(wt.to(DEV))
    g_, b_ = gamma.to(DEV), beta.to(DEV)
    # reference: normalise pass over the whole map, then the plain predictor on its slice
    norm = raw.clone()
    ws0 = ops.groupnorm_workspace(segs, G, DEV)
    ops.groupnorm_act(ops.Rows(norm), g_, b_, ops.Rows(norm), segs, G, act, ws0, 1e-5)
    y0 = torch.full((segs.rows, 8), float("nan"), device=DEV)
    ops.conv_call(ops.Rows(norm, Ct - Cin, Cin), segs, wp, ops.Rows(y0, 0, Cout), Cin=Cin, Cout=Cout, k=3, pad=1, shift=bias.to(DEV), tile=_lib.NARROW_TILE)()
    # split: statistics + affine, slice pass over the first Ct - Cin channels, gated predictor on the raw rest
    cur = raw.clone()
    ws1 = ops.groupnorm_workspace(segs, G, DEV)
    coef = torch.empty(segs.nseg * B, 2, Ct, device=DEV)
    ops.groupnorm_stats(ops.Rows(cur), g_, b_, segs, G, ws1, 1e-5, coef)
    Cs = Ct - Cin
    sl = ops.Rows(cur, 0, Cs)
    ops.coef_apply(sl, coef[:, 0, :Cs], coef[:, 1, :Cs], sl, segs, act)
    y1 = torch.full((segs.rows, 8), float("nan"), device=DEV)
    run = ops.conv_call(ops.Rows(cur, Cs, Cin), segs, wp, ops.Rows(y1, 0, Cout), Cin=Cin, Cout=Cout, k=3, pad=1, shift=bias.to(DEV), tile=_lib.NARROW_TILE,
                        gate=coef[:, 0, Cs:], gate_b=coef[:, 1, Cs:], gate_act=act)
    run()
    assert torch.equal(cur[:, :Cs], norm[:, :Cs]), "the slice pass differs from the whole-map normalise pass"
    assert torch.equal(cur[:, Cs:], raw[:, Cs:]), "the predictor's slice must stay raw"
    assert torch.equal(y1[:, :Cout], y0[:, :Cout]), "GroupNorm in the loader differs from GroupNorm in its own pass"
    assert torch.isnan(y1[:, Cout:]).all()
    # and against the definition (fp64)
    for i, ((h, w), x) in enumerate(zip(hw, xs)):
        n = F.group_norm(x.double(), G, gamma.double(), beta.double(), 1e-5)
        n = {ACT_NONE: n, ACT_RELU: F.relu(n), ACT_SILU: F.silu(n)}[act]
        r = F.conv2d(n[:, Cs:], wt.double(), bias.double(), 1, 1).float()
        g = y1[segs.m_start[i]:segs.m_start[i + 1], :Cout].cpu().reshape(B, h, w, Cout).permute(0, 3, 1, 2)
        np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=3e-5, atol=2e-5, err_msg=f"level {i}")
    with pytest.raises(Exception, match="gate_b"):          # a bare multiplicative gate has no meaning on a padded conv: clean error, no launch
        ops.conv_call(ops.Rows(cur, Cs, Cin), segs, wp, ops.Rows(y1, 0, Cout), Cin=Cin, Cout=Cout, k=3, pad=1, tile=_lib.NARROW_TILE, gate=coef[:, 0, Cs:])()


@pytest.mark.parametrize("case", [(64, 256, 64, 3 * 33 * 17, True), (64, 256, 128, 2 * 40 * 40, True), (128, 512, 128, 1000, True), (32, 64, 64, 77, False),
                                  (256, 1024, 128, 513, True)])
def test_conv1x1_back_to_back_is_bit_identical_to_two_launches(case):
    """fd_conv1x1_b2b_f32 (fd_conv_b2b.hip): a bottleneck's conv3 + BN + residual + ReLU and the next block's conv1 + BN + ReLU in one launch -- the wide
    map y is written once and never read back.  Both outputs must equal, BIT FOR BIT, two fd_conv2d launches on the workgroup-tiled kernel (same k
    order per GEMM), on row counts that are no multiple of the 64 / 32-row wave tile, channel views with NaN neighbours, with and without a residual."""
    K1, N1, N2, M, use_res = case
    gen = torch.Generator().manual_seed(K1 + N1 + N2 + M)
    segs = Segs.make(1, [(M, 1)])
    xb = torch.full((M, K1 + 8), float("nan"), device=DEV)
    xb[:, 4:4 + K1] = torch.randn(M, K1, generator=gen).to(DEV)
    x = ops.Rows(xb, 4, K1)
    w1 = (torch.randn(N1, K1, 1, 1, generator=gen) / K1 ** 0.5).to(DEV)
    w2 = (torch.randn(N2, N1, 1, 1, generator=gen) / N1 ** 0.5).to(DEV)
    s1, t1 = (torch.rand(N1, generator=gen) + 0.5).to(DEV), torch.randn(N1, generator=gen).to(DEV)
    s2, t2 = (torch.rand(N2, generator=gen) + 0.5).to(DEV), torch.randn(N2, generator=gen).to(DEV)
    rb = torch.full((M, N1 + 4), float("nan"), device=DEV)
    rb[:, :N1] = torch.randn(M, N1, generator=gen).to(DEV)
    res = ops.Rows(rb, 0, N1) if use_res else None
    # reference: two launches of the workgroup-tiled kernel
    y0, z0 = ops.new_rows(M, N1, DEV), ops.new_rows(M, N2, DEV)
    ops.conv_call(x, segs, ops.pack_conv_weight(w1), y0, Cin=K1, Cout=N1, k=1, scale=s1, shift=t1, res=res, act=ACT_RELU, tile=4)()
    ops.conv_call(y0, segs, ops.pack_conv_weight(w2), z0, Cin=N1, Cout=N2, k=1, scale=s2, shift=t2, act=ACT_RELU, tile=4)()
    yb = torch.full((M, N1 + 8), float("nan"), device=DEV)
    zb = torch.full((M, N2 + 8), float("nan"), device=DEV)
    y, z = ops.Rows(yb, 4, N1), ops.Rows(zb, 4, N2)
    ops.conv_b2b_call(x, ops.pack_conv_weight_wave(w1), y, ops.pack_conv_weight_wave(w2), z, K1=K1, N1=N1, N2=N2, scale1=s1, shift1=t1, res=res, act1=ACT_RELU,
                      scale2=s2, shift2=t2, act2=ACT_RELU)()
    assert torch.isnan(yb[:, :4]).all() and torch.isnan(yb[:, 4 + N1:]).all() and torch.isnan(zb[:, :4]).all() and torch.isnan(zb[:, 4 + N2:]).all()
    assert torch.equal(y.tensor(), y0.tensor()), float((y.tensor() - y0.tensor()).abs().max())
    assert torch.equal(z.tensor(), z0.tensor()), float((z.tensor() - z0.tensor()).abs().max())
    # and both are the convolution: fp64 reference
    yr = torch.relu(x.tensor().double() @ w1.view(N1, K1).double().t() * s1.double() + t1.double() + (res.tensor().double() if use_res else 0))
    np.testing.assert_allclose(y.tensor().cpu().numpy(), yr.float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    with pytest.raises(Exception, match="b2b"):          # N2 = 96: a clean error, no launch
        ops.conv_b2b_call(x, ops.pack_conv_weight_wave(w1), y, ops.pack_conv_weight_wave(w2), ops.new_rows(M, 96, DEV), K1=K1, N1=N1, N2=96)()


@pytest.mark.parametrize("case", [(64, 64, 256, 1, (33, 17), 8), (128, 256, 512, 2, (20, 24), 9), (256, 512, 1024, 2, (9, 7), 4), (512, 1024, 2048, 2, (5, 5), 7),
                                  (32, 96, 72, 1, (11, 13), 0)])
def test_conv1x1_with_a_k_concatenated_second_source(case):
    """fd_conv_params.x2 (the DUAL instantiations of the GEMM-addressed kernel): y = relu(o2 . Wa^T + x[::s, ::s] . Wb^T + shift) in one launch -- a
    bottleneck's conv3 and its (strided) downsample conv with the BatchNorm scales folded into the filter banks (resnet50.py:68-80) -- against fp64, on
    every tile the loader is built for, channel views with NaN neighbours, row counts that are no multiple of the tile."""
    K1, K2, N, s2, (Ho, Wo), tile = case
    B = 3
    gen = torch.Generator().manual_seed(K1 + K2 + N + tile)
    H2, W2 = s2 * Ho + (s2 - 1), s2 * Wo            # the strided source may be larger than stride * output (odd input sizes)
    a = torch.randn(B, K1, Ho, Wo, generator=gen)
    b = torch.randn(B, K2, H2, W2, generator=gen)
    wa = torch.randn(N, K1, 1, 1, generator=gen) / (K1 + K2) ** 0.5
    wb = torch.randn(N, K2, 1, 1, generator=gen) / (K1 + K2) ** 0.5
    shift = torch.randn(N, generator=gen)
    ref = torch.relu(F.conv2d(a.double(), wa.double()) + F.conv2d(b.double(), wb.double(), stride=s2)[:, :, :Ho, :Wo] + shift.double()[None, :, None, None]).float()
    segs = Segs.make(B, [(Ho, Wo)])
    ab = torch.full((segs.rows, K1 + 8), float("nan"), device=DEV)
    ab[:, 4:4 + K1] = a.permute(0, 2, 3, 1).reshape(-1, K1).to(DEV)
    bb = torch.full((B * H2 * W2, K2 + 4), float("nan"), device=DEV)
    bb[:, :K2] = b.permute(0, 2, 3, 1).reshape(-1, K2).to(DEV)
    yb = torch.full((segs.rows, N + 8), float("nan"), device=DEV)
    wp = ops.pack_conv_weight(torch.cat([wa, wb], 1).to(DEV))
    run = ops.conv_call(ops.Rows(ab, 4, K1), segs, wp, ops.Rows(yb, 4, N), Cin=K1, Cout=N, k=1, shift=shift.to(DEV), act=ACT_RELU, tile=tile,
                        x2=ops.Rows(bb, 0, K2), x2_stride=s2, x2_hw=(H2, W2))
    run()
    got = yb.cpu()
    assert torch.isnan(got[:, :4]).all() and torch.isnan(got[:, 4 + N:]).all() and not torch.isnan(got[:, 4:4 + N]).any()
    g = got[:, 4:4 + N].reshape(B, Ho, Wo, N).permute(0, 3, 1, 2)
    np.testing.assert_allclose(g.numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    with pytest.raises(Exception, match="x2"):       # split-K with a second source: a clean error
        ops.conv_call(ops.Rows(ab, 4, K1), segs, wp, ops.Rows(yb, 4, N), Cin=K1, Cout=N, k=1, tile=8, ksplit=2, workspace=torch.empty(1 << 22, device=DEV),
                      x2=ops.Rows(bb, 0, K2), x2_stride=s2, x2_hw=(H2, W2))()


@pytest.mark.parametrize("shape", [(2, 128, 160), (1, 64, 64), (3, 70, 102), (1, 262, 38), (16, 64, 96)])
def test_stem_with_fused_maxpool_is_bit_identical_to_two_launches(shape):
    """fd_stem7x7_pool_nhwc4: conv1 + bn1 + relu + maxpool(3, 2, 1) in one launch (windows inside a workgroup's tiles stored once, windows across workgroup
    borders combined by integer atomicMax on a zero-filled map) == the stem launch followed by the max-pool launch, BIT FOR BIT -- strips of 4 tiles that
    end inside the image, odd output sizes (partial tiles, floor / ceil of the pool geometry), several images, a channel view with NaN neighbours."""
    B, H, W = shape
    gen = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(B, 3, H, W, generator=gen)
    w = torch.randn(64, 3, 7, 7, generator=gen) / 147 ** 0.5
    sc, sf = (torch.rand(64, generator=gen) + 0.5).to(DEV), (torch.randn(64, generator=gen) * 0.5).to(DEV)
    x4 = ops.new_rows(B * H * W, 4, DEV)
    ops.nchw3_to_nhwc4(x.to(DEV), x4.buf)
    wp = ops.pack_stem7_weight(w.to(DEV))
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    y1 = ops.new_rows(B * H1 * W1, 64, DEV)
    y2 = ops.new_rows(B * H2 * W2, 64, DEV)
    ops.stem7x7(x4, wp, y1, B, H, W, sc, sf, ACT_RELU)
    ops.maxpool(y1, y2, B, H1, W1, 3, 2, 1)
    zb = torch.full((B * H2 * W2, 72), float("nan"), device=DEV)
    z = ops.Rows(zb, 4, 64)
    ops.stem7x7_pool(x4, wp, z, B, H, W, sc, sf)
    assert torch.isnan(zb[:, :4]).all() and torch.isnan(zb[:, 68:]).all(), "wrote outside its channel view"
    assert torch.equal(z.tensor(), y2.tensor()), float((z.tensor() - y2.tensor()).abs().max())
    first = z.tensor().clone()
    ops.stem7x7_pool(x4, wp, z, B, H, W, sc, sf)
    assert torch.equal(z.tensor(), first)                      # max is order-independent: the atomics leave no run-to-run difference
    # and it is the reference arithmetic: conv -> BN fold -> ReLU -> max_pool2d
    ref = F.max_pool2d(torch.relu(F.conv2d(x.double(), w.double(), None, 2, 3) * sc.cpu().double()[None, :, None, None] + sf.cpu().double()[None, :, None, None]), 3, 2, 1)
    got = z.tensor().cpu().reshape(B, H2, W2, 64).permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), ref.float().numpy(), rtol=2e-5, atol=2e-5)
    # fd_stem7x7_nchw3: both kernels with the reference's own [N, 3, H, W] tensor read by the patch loader (no [N][H][W][4] copy) -- the same values in the
    # same order: bit-identical; the input sits inside a larger allocation whose neighbours are NaN
    pad = torch.full((B * 3 * H * W + 32,), float("nan"), device=DEV)
    xn = pad[16:16 + B * 3 * H * W].view(B, 3, H, W)
    xn.copy_(x.to(DEV))
    zb2 = torch.full((B * H2 * W2, 72), float("nan"), device=DEV)
    ops.stem7x7_nchw(xn, wp, ops.Rows(zb2, 4, 64), sc, sf, pool=True)
    assert torch.equal(zb2.nan_to_num(7.0), zb.nan_to_num(7.0))
    y1n = ops.new_rows(B * H1 * W1, 64, DEV)
    ops.stem7x7_nchw(xn, wp, y1n, sc, sf, ACT_RELU)
    assert torch.equal(y1n.tensor(), y1.tensor())
    with pytest.raises(Exception, match="contiguous fp32"):
        ops.stem7x7_nchw(xn.permute(0, 1, 3, 2), wp, y1n, sc, sf, ACT_RELU)


def test_conv3x3_winograd_f4x4_at_the_start_of_an_allocation():
    """VERDICT r3 hygiene item 15: fd_conv_wino4.hip builds its input buffer resource ONE patch pixel before the tensor (so that the six columns of a patch
    row are one per-lane offset plus non-negative scalar offsets) and relies on per-lane masking of column 0 at w0 = 0.  Here the input view has x_co = 0 and
    starts exactly at the start of a fresh, page-aligned device allocation: nothing below it may be touched (the page below need not be mapped), and the
    first pixels' results must be right."""
    from pytorch_object_detection_amd import _lib
    torch.cuda.empty_cache()
    Cin, Cout, hw, B = 64, 64, [(24, 20)], 2
    segs = Segs.make(B, hw)
    flat = torch.empty(64 << 20, dtype=torch.float32, device=DEV)          # a fresh 256 MB segment of its own: the block starts the segment
    if flat.data_ptr() % 4096:
        pytest.skip("the allocator did not hand out a page-aligned block")
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, *hw[0], generator=gen)
    xr = flat[:segs.rows * Cin].view(segs.rows, Cin)
    xr.copy_(x.permute(0, 2, 3, 1).reshape(-1, Cin).to(DEV))
    assert xr.data_ptr() == flat.data_ptr()
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    y = ops.new_rows(segs.rows, Cout, DEV)
    for dil in (1, 2):
        ops.conv_call(ops.Rows(xr), segs, ops.pack_conv_weight_wino4(wt.to(DEV)), y, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=_lib.WINO4_TILE)()
        ref = F.conv2d(x.double(), wt.double(), None, 1, dil, dil).float()
        got = y.tensor().cpu().reshape(B, *hw[0], Cout).permute(0, 3, 1, 2)
        scale = float(ref.abs().max()) + 1.0
        assert float((got - ref).abs().max()) < 4e-5 * scale
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(64, 128, [(40, 40), (20, 20), (10, 10), (5, 5), (3, 3)], 3, 1), (32, 80, [(48, 36)], 2, 2), (64, 64, [(24, 20)], 1, 1)])
def test_conv3x3_winograd_f4x4_as_slices_of_its_grid(case):
    """fd_conv_params.wg_first / wg_count: an F(4x4) layer launched as several slices of its workgroup grid (the head tower as whole rounds on 256 CUs + a
    tail launch, engine.TOWER_TAIL_SPLIT).  Any partition of [0, fd_conv_workgroups) at multiples of 8 must give, BIT FOR BIT, the one-launch result, every
    slice must write only its own tiles, the non-empty workgroup counts of the slices must add up, and ranges outside the grid / off an XCD boundary /
    on another tile are clean errors."""
    from pytorch_object_detection_amd import _lib
    Cin, Cout, hw, B, dil = case
    gen = torch.Generator().manual_seed(Cin + Cout + len(hw) + dil)
    segs = Segs.make(B, hw)
    x = torch.randn(segs.rows, Cin, generator=gen).to(DEV)
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=gen).to(DEV)
    wp = ops.pack_conv_weight_wino4(wt.to(DEV))
    y0 = ops.new_rows(segs.rows, Cout, DEV)
    whole = ops.conv_call(ops.Rows(x), segs, wp, y0, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, shift=bias, act=ACT_RELU, tile=_lib.WINO4_TILE)
    whole()
    total, live = ops.conv_workgroups(whole)
    assert total % 8 == 0 and 0 < live <= total
    yb = torch.full((segs.rows, Cout), float("nan"), device=DEV)
    call = ops.conv_call(ops.Rows(x), segs, wp, ops.Rows(yb), Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, shift=bias, act=ACT_RELU, tile=_lib.WINO4_TILE)
    cuts = sorted({0, total, (total // 3) // 8 * 8, (2 * total // 3) // 8 * 8, max(0, total - 8)})
    parts = [ops.conv_wg_slice(call, a, b - a, tag=i & 1) for i, (a, b) in enumerate(zip(cuts, cuts[1:])) if b > a]
    assert sum(ops.conv_workgroups(q)[1] for q in parts) == live
    seen = torch.zeros(segs.rows, dtype=torch.bool, device=DEV)
    for q in reversed(parts):                    # (any order)
        q()
        now = ~torch.isnan(yb).any(1)
        assert bool((now | ~seen).all())           # nothing a previous slice wrote is touched again ...
        assert torch.equal(yb[seen], y0.tensor()[seen])
        seen = now
    assert bool(seen.all()) and torch.equal(yb, y0.tensor())
    for first, count in ((4, 8), (0, total + 8), (total, 8), (-8, 8)):
        with pytest.raises(Exception, match="wg_"):
            ops.conv_wg_slice(call, first, count)()
    direct = ops.conv_call(ops.Rows(x), segs, ops.pack_conv_weight(wt.to(DEV)), ops.Rows(yb), Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=4)
    with pytest.raises(Exception, match="WINOGRAD4"):
        ops.conv_wg_slice(direct, 0, 8)()


@pytest.mark.parametrize("case", [(64, 128, [(40, 40), (20, 20), (10, 10), (5, 5), (3, 3)], 3, 1, 64), (32, 80, [(48, 36)], 2, 2, 32), (64, 64, [(24, 20)], 1, 1, 8),
                                  (256, 80, [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)], 4, 1, 256), (128, 128, [(80, 80)], 16, 1, 256), (64, 192, [(32, 32)], 2, 2, 120), (64, 64, [(64, 64)], 33, 1, 256), (64, 64, [(64, 64)], 32, 1, 256)])
def test_conv3x3_winograd_f4x4_as_a_persistent_stream_k_grid(case):
    """fd_conv_params.sk_wgs: the F(4x4) layer as a persistent grid -- whole rounds of items round-robin, the last (partial) round's (tile block, chunk) units shared
    evenly by the workgroups (the head tower and cls_logits of HISFcos.py:196-209 on 256 CUs without a last, mostly idle round; cases: pyramids, dilation 2, XCDs without
    tiles, a remainder with fewer units than workgroups, no remainder at all).  Items that no range boundary cuts must equal the plain launch BIT FOR BIT; a cut item is
    the sum of its parts' output transforms (fixed order): within the F(4x4) bound of the fp64 reference and of the plain launch, identical from run to run; the flags are
    zero again after every launch (the workspace is reused without clearing); channel views with NaN neighbours; bad grids / workspaces are clean errors."""
    from pytorch_object_detection_amd import _lib
    Cin, Cout, hw, B, dil, wgs = case
    gen = torch.Generator().manual_seed(Cin + Cout + len(hw) + dil)
    segs = Segs.make(B, hw)
    x = torch.randn(segs.rows, Cin, generator=gen)
    wt = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    sc, bias = (torch.rand(Cout, generator=gen) + 0.5).to(DEV), torch.randn(Cout, generator=gen).to(DEV)
    res = torch.randn(segs.rows, Cout, generator=gen).to(DEV)
    wp = ops.pack_conv_weight_wino4(wt.to(DEV))
    xb = torch.full((segs.rows, Cin + 8), float("nan"), device=DEV)
    xb[:, 4:4 + Cin] = x.to(DEV)
    kw = dict(Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, scale=sc, shift=bias, res=ops.Rows(res), act=ACT_RELU, tile=_lib.WINO4_TILE)
    y0 = ops.new_rows(segs.rows, Cout, DEV)
    ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y0, **kw)()
    ws = ops.sk_workspace(wgs, DEV)
    yb = torch.full((segs.rows, Cout + 8), float("nan"), device=DEV)
    run = ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb, 4, Cout), sk_wgs=wgs, workspace=ws, **kw)
    run()
    torch.cuda.synchronize()
    assert int(ws[:2048].view(torch.int32).abs().sum()) == 0, "a slot flag / queue head was left non-zero"
    if wgs > 16:                   # launches with different grids share one workspace (a fixed header): a smaller grid in between must not disturb the next launch
        ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb, 4, Cout), sk_wgs=wgs - 16, workspace=ws, **kw)()
        yb.fill_(float("nan"))
        run()
        torch.cuda.synchronize()
        assert int(ws[:2048].view(torch.int32).abs().sum()) == 0
    assert torch.isnan(yb[:, :4]).all() and torch.isnan(yb[:, 4 + Cout:]).all(), "wrote outside its channel view"
    got = yb[:, 4:4 + Cout].clone()
    assert not torch.isnan(got).any(), "an output pixel was never written"
    same = (got == y0.tensor()).all(1)
    scale = float(y0.tensor().abs().max()) + 1.0
    assert float((got - y0.tensor()).abs().max()) < 1e-5 * scale            # cut items: another summation order of the same products
    # expected cut structure: per XCD, the interior range boundaries that do not fall on an item boundary cut one item each
    NC, T = Cin // 8, sum(B * dil * dil * (((h + dil - 1) // dil + 3) // 4) * (((w + dil - 1) // dil + 3) // 4) for h, w in hw)
    mtiles, ntiles = (T + 31) // 32, (Cout + 63) // 64
    mt_per, wpx, cut_items = (mtiles + 7) // 8, wgs // 8, 0
    for xcd in range(8):           # per XCD (mtiles / 8 M tiles or one more): only the items % wpx last items of its queue may be cut into pieces
        cnt = mtiles // 8 + (1 if xcd < mtiles % 8 else 0)
        cut_items += cnt * ntiles % wpx
    rows_per_item_max = 32 * 16
    assert int((~same).sum()) <= cut_items * rows_per_item_max, (int((~same).sum()), cut_items)
    if cut_items == 0:
        assert bool(same.all())
    # per-level fp64 reference
    for i, (h, w) in enumerate(hw):
        lo, hi = segs.m_start[i], segs.m_start[i + 1]
        xi = x[lo:hi].reshape(B, h, w, Cin).permute(0, 3, 1, 2).double()
        r = F.conv2d(xi, wt.double(), None, 1, dil, dil) * sc.cpu().double()[None, :, None, None] + bias.cpu().double()[None, :, None, None]
        r = F.relu(r + res[lo:hi].cpu().double().reshape(B, h, w, Cout).permute(0, 3, 1, 2)).float()
        g = got[lo:hi].cpu().reshape(B, h, w, Cout).permute(0, 3, 1, 2)
        s_ = float(r.abs().max()) + 1.0
        assert float((g - r).abs().max()) < 4e-5 * s_
    for _ in range(3):                         # same workspace, no clearing: deterministic
        yb.fill_(float("nan"))
        run()
        assert torch.equal(yb[:, 4:4 + Cout], got)
    for bad in (dict(sk_wgs=12, workspace=ws), dict(sk_wgs=wgs, workspace=None), dict(sk_wgs=wgs, workspace=ws[:1024]), dict(sk_wgs=wgs, workspace=ws, ksplit=2),
                dict(sk_wgs=2048, workspace=ws)):
        with pytest.raises(Exception, match="sk_wgs"):
            ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb, 4, Cout), **kw, **bad)()
    direct = ops.conv_call(ops.Rows(xb, 4, Cin), segs, ops.pack_conv_weight(wt.to(DEV)), ops.Rows(yb, 4, Cout), Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=4, sk_wgs=8,
                           workspace=ws)
    with pytest.raises(Exception, match="WINOGRAD4"):
        direct()


@pytest.mark.parametrize("case", [(1024, 256, 2, 40, 40, 9), (512, 256, 1, 80, 80, 8), (64, 96, 3, 6, 10, 4), (128, 64, 2, 14, 2, 0)])
def test_conv1x1_with_the_upsampled_coarser_level_added_in_its_epilogue(case):
    """fd_conv_params.res_mode 2: an FPN lateral -- relu(bn(conv1x1(c))) + Upsample(x2, nearest)(coarser) (HISFcos.py:155-165; without the activation: Fcos.py:77-91)
    -- as ONE launch: the epilogue reads the half-resolution map at (i / 2, j / 2) and adds it AFTER the activation.  Must equal, BIT FOR BIT, the conv launch
    followed by fd_upsample2x_add_nhwc (same two addends per element), and F.interpolate + F.conv2d in fp64; channel views with NaN neighbours; odd sizes / other
    tiles / split-K are clean errors."""
    Cin, Cout, B, H, W, tile = case
    gen = torch.Generator().manual_seed(Cin + Cout + H)
    segs = Segs.make(B, [(H, W)])
    x = torch.randn(B, Cin, H, W, generator=gen)
    up = torch.randn(B, Cout, H // 2, W // 2, generator=gen)
    wt = torch.randn(Cout, Cin, 1, 1, generator=gen) / Cin ** 0.5
    sc, sf = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen) * 0.3
    xb = torch.full((segs.rows, Cin + 8), float("nan"), device=DEV)
    xb[:, 4:4 + Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin).to(DEV)
    ub = torch.full((B * (H // 2) * (W // 2), Cout + 4), float("nan"), device=DEV)
    ub[:, :Cout] = up.permute(0, 2, 3, 1).reshape(-1, Cout).to(DEV)
    wp = ops.pack_conv_weight(wt.to(DEV))
    # two launches
    y0 = ops.new_rows(segs.rows, Cout, DEV)
    ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, y0, Cin=Cin, Cout=Cout, k=1, scale=sc.to(DEV), shift=sf.to(DEV), act=ACT_RELU, tile=tile)()
    ops.upsample2x_add(ops.Rows(ub, 0, Cout), y0, y0, B, H // 2, W // 2)
    # one launch
    yb = torch.full((segs.rows, Cout + 8), float("nan"), device=DEV)
    ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb, 4, Cout), Cin=Cin, Cout=Cout, k=1, scale=sc.to(DEV), shift=sf.to(DEV), act=ACT_RELU, tile=tile,
                  res=ops.Rows(ub, 0, Cout), res_up=True)()
    assert torch.isnan(yb[:, :4]).all() and torch.isnan(yb[:, 4 + Cout:]).all(), "wrote outside its channel view"
    assert torch.equal(yb[:, 4:4 + Cout], y0.tensor()), float((yb[:, 4:4 + Cout] - y0.tensor()).abs().max())
    ref = (F.relu(F.conv2d(x.double(), wt.double()) * sc.double()[None, :, None, None] + sf.double()[None, :, None, None])
           + F.interpolate(up.double(), scale_factor=2, mode="nearest")).float()
    got = yb[:, 4:4 + Cout].cpu().reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    for bad in (dict(tile=1), dict(tile=9, ksplit=2, workspace=torch.empty(1 << 22, device=DEV))):
        with pytest.raises(Exception, match="res_mode 2"):
            ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb, 4, Cout), Cin=Cin, Cout=Cout, k=1, res=ops.Rows(ub, 0, Cout), res_up=True, **bad)()
    # ... and so is every combination whose kernel is not an RUP instantiation (it would read the quarter-size map at all M rows): rejected in front of the
    # library's early returns (gate, f16 / f16x3, gn_stats, tag 1), nothing is launched and the output keeps its contents
    before = yb.clone()
    gate = torch.ones(B, Cin, device=DEV)
    for bad in (dict(gate=gate), dict(precision=1), dict(precision=2), dict(tag=1), dict(tag=1, tile=8),
                dict(gn_stats=torch.zeros(segs.rows, 32, 2, device=DEV), gn_groups=32)):
        with pytest.raises(Exception, match="res_mode 2"):
            ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb, 4, Cout), Cin=Cin, Cout=Cout, k=1, res=ops.Rows(ub, 0, Cout), res_up=True, **bad)()
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(yb, nan=-7.0), torch.nan_to_num(before, nan=-7.0))
    with pytest.raises(Exception, match="res_up"):           # a residual of the wrong size never reaches the library
        ops.conv_call(ops.Rows(xb, 4, Cin), segs, wp, ops.Rows(yb, 4, Cout), Cin=Cin, Cout=Cout, k=1, res=y0, res_up=True)
