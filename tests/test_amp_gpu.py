"""The reference's TRAINING arithmetic is AMP fp16 (train.py:33 amp_enabled = True, :175-181 autocast + GradScaler; config/main.yaml:5).
Under torch.autocast(float16) the HIP nodes run dense convolutions -- forward, data gradient -- with f16 operands on
v_mfma_f32_32x32x16_f16 and fp32 accumulation (FD_PREC_F16), everything else in fp32 as autocast does.  Oracle: the same op on the CPU
with its operands rounded to f16 and fp32 accumulation; the tolerance covers summation order only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pytorch_object_detection_amd import _lib, ops, train_ops as T
from pytorch_object_detection_amd._lib import ACT_NONE, ACT_RELU, ACT_SILU, Segs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def h(t):
    return t.half().float()


CASES = [
    # Cin, Cout, k, stride, pad, dil, hw (pyramid), act, residual
    (64, 256, 1, 1, 0, 1, [(20, 24)], ACT_RELU, True),
    (256, 64, 1, 1, 0, 1, [(20, 24)], ACT_RELU, False),
    (128, 128, 3, 2, 1, 1, [(21, 13)], ACT_RELU, False),
    (256, 256, 3, 1, 2, 2, [(12, 12)], ACT_NONE, False),
    (256, 512, 3, 1, 1, 1, [(10, 12), (5, 6), (3, 3), (1, 2)], ACT_NONE, False),      # the head tower over a pyramid
    (256, 80, 3, 1, 1, 1, [(9, 9), (4, 5)], ACT_NONE, False),
    (2048, 256, 1, 1, 0, 1, [(6, 7)], ACT_RELU, False),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_f16_operands_fp32_accumulate(case):
    Cin, Cout, k, stride, pad, dil, hw, act, use_res = case
    gen = torch.Generator().manual_seed(Cin + Cout + k)
    B = 2
    xs = [torch.randn(B, Cin, a, b, generator=gen) for a, b in hw]
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    scale, shift = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen)
    segs = Segs.make(B, hw)
    so = ops.conv_out_segs(segs, k, stride, pad, dil)
    rs = [torch.randn(B, Cout, a, b, generator=gen) for a, b in so.level_hw()]
    xr = ops.Rows(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(DEV))
    rr = ops.Rows(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in rs]).to(DEV)) if use_res else None
    y = ops.new_rows(so.rows, Cout, DEV)
    wp = ops.pack_conv_weight_hip(w.to(DEV), f16=True)
    ops.conv_call(xr, segs, wp, y, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale.to(DEV), shift=shift.to(DEV), res=rr, act=act,
                  precision=_lib.PREC_F16)()
    got = y.tensor().cpu()
    for lv, ((a, b), x, r) in enumerate(zip(so.level_hw(), xs, rs)):
        ref = F.conv2d(h(x), h(w), None, stride, pad, dil) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if use_res:
            ref = ref + r
        if act == ACT_RELU:
            ref = F.relu(ref)
        g = got[so.m_start[lv]:so.m_start[lv + 1]].reshape(B, a, b, Cout).permute(0, 3, 1, 2)
        np.testing.assert_allclose(g.numpy(), ref.numpy(), atol=2e-5, rtol=2e-5, err_msg=f"level {lv}")
    # and it is NOT the fp32 result (the operands really were rounded)
    ref32 = F.conv2d(xs[0], w, None, stride, pad, dil)
    assert float((F.conv2d(h(xs[0]), h(w), None, stride, pad, dil) - ref32).abs().max()) > 1e-4


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("io", [(True, True, True), (True, False, False), (False, True, True)])
def test_conv_f16_activation_maps_in_hbm(case, io):
    """fd_conv_params.io_f16: under autocast the reference's convolutions read and write fp16 TENSORS (train.py:175-181), so AMP activations can live in HBM as f16:
    the loader fetches f16 and feeds the MFMA without conversion, the epilogue rounds the fp32 result once.  Must equal, BIT FOR BIT, the fp32-map launch of the
    same kernel on the same (already f16-valued) inputs followed by one rounding of its output -- for f16 inputs, outputs and residuals in any combination, over
    pyramids / strides / dilation, with channel views whose neighbours are NaN, with the residual as ReLU mask, and through split-K."""
    Cin, Cout, k, stride, pad, dil, hw, act, use_res = case
    x16, y16, r16 = io
    gen = torch.Generator().manual_seed(Cin + Cout + k + 7)
    B = 2
    segs = Segs.make(B, hw)
    so = ops.conv_out_segs(segs, k, stride, pad, dil)
    x = h(torch.randn(segs.rows, Cin, generator=gen)).to(DEV)                      # f16-representable values in both forms
    r = h(torch.randn(so.rows, Cout, generator=gen)).to(DEV)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    scale, shift = (torch.rand(Cout, generator=gen) + 0.5).to(DEV), torch.randn(Cout, generator=gen).to(DEV)
    wp = ops.pack_conv_weight_hip(w.to(DEV), f16=True)
    for res_mask in ((False, True) if use_res else (False,)):
        kw = dict(Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale, shift=shift, act=act, precision=_lib.PREC_F16, res_mask=res_mask)
        y0 = ops.new_rows(so.rows, Cout, DEV)
        ops.conv_call(ops.Rows(x), segs, wp, y0, res=ops.Rows(r) if use_res else None, **kw)()
        want = y0.tensor().half() if y16 else y0.tensor()

        def buf(t, f16, co, C_):
            b = torch.full((t.shape[0], C_ + 12), float("nan"), dtype=torch.float16 if f16 else torch.float32, device=DEV)
            b[:, co:co + C_] = t
            return b
        xb, rb = buf(x, x16, 8, Cin), buf(r, r16, 4, Cout)
        yb = torch.full((so.rows, Cout + 12), float("nan"), dtype=torch.float16 if y16 else torch.float32, device=DEV)
        for ks in (1, 2):
            if ks > 1 and (Cin // 32) * k * k < 4:
                continue
            yb.fill_(float("nan"))
            ws = torch.empty(ks * so.rows * Cout + 64, device=DEV) if ks > 1 else None
            ops.conv_call(ops.Rows(xb, 8, Cin), segs, wp, ops.Rows(yb, 4, Cout), res=ops.Rows(rb, 4, Cout) if use_res else None, ksplit=ks, workspace=ws, tile=2, **kw)()
            assert torch.isnan(yb[:, :4]).all() and torch.isnan(yb[:, 4 + Cout:]).all(), "wrote outside its channel view"
            got = yb[:, 4:4 + Cout]
            if ks == 1:
                ops.conv_call(ops.Rows(x), segs, wp, y0, res=ops.Rows(r) if use_res else None, tile=2, **kw)()      # same tile: same summation order
                want = y0.tensor().half() if y16 else y0.tensor()
                assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
            else:
                np.testing.assert_allclose(got.float().cpu().numpy(), want.float().cpu().numpy(), atol=2e-3 if y16 else 2e-5, rtol=2e-3 if y16 else 2e-5)
    with pytest.raises(Exception, match="FD_PREC_F16"):
        ops.conv_call(ops.Rows(x.half()), segs, ops.pack_conv_weight(w.to(DEV)), ops.new_rows(so.rows, Cout, DEV), Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil)


K64_CASES = [
    # Cin, Cout, k, stride, pad, dil, hw (pyramid), act, residual
    (64, 256, 1, 1, 0, 1, [(20, 24)], ACT_RELU, True),
    (256, 64, 1, 1, 0, 1, [(20, 24)], ACT_RELU, False),
    (128, 128, 3, 2, 1, 1, [(21, 13)], ACT_RELU, False),
    (256, 256, 3, 1, 2, 2, [(12, 12)], ACT_NONE, False),
    (256, 512, 3, 1, 1, 1, [(10, 12), (5, 6), (3, 3), (1, 2)], ACT_NONE, False),
    (512, 128, 1, 2, 0, 1, [(14, 10)], ACT_NONE, False),
    (2048, 256, 1, 1, 0, 1, [(6, 7)], ACT_RELU, False),
    (64, 64, 3, 1, 1, 1, [(40, 36)], ACT_RELU, True),           # enough rows for the 128-row tiles
    (128, 512, 1, 1, 0, 1, [(64, 48)], ACT_RELU, True),         # ... and the 128 x 128 one
    (64, 20, 3, 1, 1, 1, [(9, 11), (5, 6)], ACT_NONE, True),    # Cout % 8 != 0: the epilogue's 8-byte accesses (general instantiation)
    (128, 72, 1, 1, 0, 1, [(30, 34)], ACT_SILU, False),         # SiLU
]


@pytest.mark.parametrize("case", K64_CASES)
@pytest.mark.parametrize("io", [(False, False, False), (True, True, True), (True, False, False), (False, True, True)])
def test_conv_f16_on_k_tiles_of_64_channels(case, io):
    """FD_TILE_F16K64 (fd_conv_f16.hip): the AMP step's conv kernel -- f16 operands, fp32 accumulation, K-tiles of 64 channels, activation maps in HBM as f16 or fp32.
    Oracle: the same op on the CPU with operands rounded to f16 and fp32 accumulation (tolerance = summation order); an f16 output is the fp32 result of the same launch
    rounded once (bit-exact between the two output types); channel views with NaN neighbours; residual as addend and as ReLU mask."""
    Cin, Cout, k, stride, pad, dil, hw, act, use_res = case
    x16, y16, r16 = io
    gen = torch.Generator().manual_seed(Cin + Cout + k + 11)
    B = 2
    xs = [h(torch.randn(B, Cin, a, b, generator=gen)) for a, b in hw]
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    scale, shift = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen)
    segs = Segs.make(B, hw)
    so = ops.conv_out_segs(segs, k, stride, pad, dil)
    rs = [h(torch.randn(B, Cout, a, b, generator=gen)) for a, b in so.level_hw()]
    x = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(DEV)
    r = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in rs]).to(DEV)
    wp = ops.pack_conv_weight_f16k64(w.to(DEV))

    def buf(t, f16, co, C_):
        b = torch.full((t.shape[0], C_ + 16), float("nan"), dtype=torch.float16 if f16 else torch.float32, device=DEV)
        b[:, co:co + C_] = t
        return b
    for res_mask in ((False, True) if use_res else (False,)):
        xb, rb = buf(x, x16, 8, Cin), buf(r, r16, 4, Cout)
        yb = torch.full((so.rows, Cout + 16), float("nan"), dtype=torch.float16 if y16 else torch.float32, device=DEV)
        kw = dict(Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale.to(DEV), shift=shift.to(DEV), act=act, precision=_lib.PREC_F16,
                  res_mask=res_mask, tile=_lib.F16K64_TILE)
        ops.conv_call(ops.Rows(xb, 8, Cin), segs, wp, ops.Rows(yb, 4, Cout), res=ops.Rows(rb, 4, Cout) if use_res else None, **kw)()
        assert torch.isnan(yb[:, :4]).all() and torch.isnan(yb[:, 4 + Cout:]).all(), "wrote outside its channel view"
        got = yb[:, 4:4 + Cout].float().cpu()
        assert not torch.isnan(got).any()
        for lv, ((a, b), xl, rl) in enumerate(zip(so.level_hw(), xs, rs)):
            ref = F.conv2d(xl, h(w), None, stride, pad, dil) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            if use_res:
                ref = torch.where(rl > 0, ref, torch.zeros_like(ref)) if res_mask else ref + rl
            if act == ACT_RELU:
                ref = F.relu(ref)
            elif act == ACT_SILU:
                ref = F.silu(ref)
            g = got[so.m_start[lv]:so.m_start[lv + 1]].reshape(B, a, b, Cout).permute(0, 3, 1, 2)
            tol = 2e-3 if y16 else (1e-4 if act == ACT_SILU else 2e-5)      # (SiLU on v_exp_f32 / v_rcp_f32)
            np.testing.assert_allclose(g.numpy(), ref.numpy(), atol=tol, rtol=tol, err_msg=f"level {lv}")
        if y16:      # ... and exactly the fp32-output launch, rounded once
            y32 = torch.empty(so.rows, Cout, device=DEV)
            ops.conv_call(ops.Rows(xb, 8, Cin), segs, wp, ops.Rows(y32), res=ops.Rows(rb, 4, Cout) if use_res else None, **kw)()
            assert torch.equal(yb[:, 4:4 + Cout], y32.half())
    with pytest.raises(Exception, match="F16K64"):
        ops.conv_call(ops.Rows(x), segs, wp, ops.new_rows(so.rows, Cout, DEV), Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, tile=_lib.F16K64_TILE)()


def test_pack_f16_pair_format():
    gen = torch.Generator().manual_seed(1)
    w = torch.randn(96, 64, 3, 3, generator=gen)
    a = ops.pack_conv_weight_hip(w.to(DEV), f16=True).view(torch.float16).cpu()
    b = ops.pack_conv_weight_f16x3(w.to(DEV)).cpu().reshape(-1)
    assert torch.equal(a.reshape(-1), b)
    sc = torch.rand(96, generator=gen) + 0.5
    a2 = ops.pack_conv_weight_hip(w.to(DEV), sc.to(DEV), dgrad=True, f16=True).view(torch.float16).cpu().reshape(-1)
    wd = (w * sc.view(-1, 1, 1, 1)).flip(2, 3).transpose(0, 1).contiguous()
    assert torch.equal(a2, ops.pack_conv_weight_f16x3(wd.to(DEV)).cpu().reshape(-1))


@pytest.mark.parametrize("k,stride", [(1, 1), (3, 1), (3, 2)])
def test_autocast_conv_node_forward_and_data_gradient(k, stride):
    """train_ops.conv_bn_act under torch.autocast(float16): forward = conv of f16-rounded (x, w); data gradient = transposed conv of
    f16-rounded (dy, w * bn_scale); weight gradient against autograd of the same rounded operands."""
    gen = torch.Generator().manual_seed(10 * k + stride)
    B, Cin, Cout, H, W = 2, 64, 128, 14, 12
    conv = torch.nn.Conv2d(Cin, Cout, k, stride, k // 2, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) / np.sqrt(Cin * k * k))
    x = torch.randn(B, Cin, H, W, generator=gen)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    dy = torch.randn(B, Cout, Ho, Wo, generator=gen)
    xd = x.to(DEV).to(memory_format=torch.channels_last).requires_grad_(True)
    convd = torch.nn.Conv2d(Cin, Cout, k, stride, k // 2, bias=False).to(DEV)
    with torch.no_grad():
        convd.weight.copy_(conv.weight)
    with torch.autocast("cuda", dtype=torch.float16):
        assert T.amp_prec() == _lib.PREC_F16
        y = T.conv_bn_act(convd, None, xd, ACT_NONE)
    assert y.dtype == torch.float32
    y.backward(dy.to(DEV))
    xr = h(x).requires_grad_(True)
    wr = h(conv.weight.detach()).requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride, k // 2)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5, rtol=2e-5)
    gx_ref = torch.nn.grad.conv2d_input(x.shape, h(conv.weight.detach()), h(dy), stride, k // 2)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), gx_ref.numpy(), atol=3e-5, rtol=3e-5)
    gw = convd.weight.grad.cpu()
    gw_exact = torch.nn.grad.conv2d_weight(x, conv.weight.shape, dy, stride, k // 2)
    gw_f16 = torch.nn.grad.conv2d_weight(h(x), conv.weight.shape, h(dy), stride, k // 2)
    err_exact, err_f16 = float((gw - gw_exact).abs().max()), float((gw - gw_f16).abs().max())
    assert err_f16 < 5e-5 * float(gw_exact.abs().max()) + 1e-5 and err_f16 < err_exact, (err_exact, err_f16)      # the f16-operand weight gradient


@pytest.mark.parametrize("cfg", [(256, 64, 1, False, torch.float32), (256, 128, 2, True, torch.float32), (512, 128, 1, False, torch.float16), (512, 256, 2, True, torch.float16)])
def test_bottleneck_node_keeps_its_maps_in_f16_under_autocast(cfg):
    """train_ops.AMP_F16_STORE: under autocast a ResNet bottleneck keeps y1, y2, the identity branch, its output and every gradient that flows back through it in HBM
    as f16 (the reference's autocast convolutions produce fp16 tensors, train.py:175-181).  Against the SAME node with fp32 maps (round 3's AMP step: identical f16
    products, activations rounded on their way to LDS instead of once when stored): outputs within f16 rounding, input and weight gradients with cosine > 0.999 and fewer than 1 % of the
    elements off by more than 2 % of the maximum (ReLU-mask flips, see below); f16 and fp32 inputs; the dtype contract (f16 out, gradient of the input in the input's own type)."""
    from pytorch_object_detection_amd.model.backbone.resnet50 import _Bottleneck
    Cin, P, stride, ds, xdt = cfg
    torch.manual_seed(Cin + P + stride)
    blk = _Bottleneck(Cin, P, stride, ds).to(DEV)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            m.weight.requires_grad_(False); m.bias.requires_grad_(False)
    blk.eval()                      # frozen BatchNorm (HISFcos.py:57-68)
    B, H, W = 2, 18, 14
    x0 = torch.randn(B, Cin, H, W, device=DEV).to(memory_format=torch.channels_last)
    gy = torch.randn(B, 4 * P, (H - 1) // stride + 1, (W - 1) // stride + 1, device=DEV).to(memory_format=torch.channels_last) * 64.0     # (a GradScaler-sized gradient)
    res = {}
    for store in (False, True):
        T.AMP_F16_STORE = store
        try:
            blk.zero_grad(set_to_none=True)
            x = x0.to(xdt).detach().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.float16):
                y = T.bottleneck(blk, x)
            assert y.dtype == (torch.float16 if store else torch.float32)
            y.backward(gy.to(y.dtype))
            assert x.grad.dtype == xdt
            res[store] = (y.detach().float(), x.grad.float(), [p.grad.float().clone() for p in blk.parameters() if p.requires_grad])
        finally:
            T.AMP_F16_STORE = True
    y0, gx0, gw0 = res[False]
    y1, gx1, gw1 = res[True]
    assert float((y1 - y0).abs().max()) <= 2e-3 * (float(y0.abs().max()) + 1.0)
    # The identity branch reaches the block's ReLU rounded to f16 (as in torch's autocast) instead of in fp32: a pre-activation within 1e-3 of zero can land on the
    # other side, and one flipped mask element moves a whole row of the gradients upstream by |dy| * w (measured: 59 of 32 256 elements of one layer's gradient,
    # all in one pixel row).  So: direction and bulk agreement, not an element-wise bound.
    for a, b in [(gx1, gx0)] + list(zip(gw1, gw0)):
        assert float(F.cosine_similarity(a.flatten(), b.flatten(), dim=0)) > 0.999
        bad = ((a - b).abs() > 2e-2 * float(b.abs().max())).float().mean()
        assert float(bad) < 0.01, float(bad)
    assert T.STATS.get("stock_fallbacks", 0) == 0


def test_amp_train_step_runs_on_f16_mfma_and_tracks_the_fp32_step():
    """One HISFCOS-R50 step under autocast + GradScaler (train.py:175-181): loss within 1 % of the fp32 step, finite gradients for every
    trainable parameter, no stock fallback (FD_STRICT), and the conv launches really took the f16 path."""
    from pytorch_object_detection_amd.model.loss import FCOSLoss
    from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
    from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
    torch.manual_seed(0)
    model = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).to(DEV).train()
    x = torch.randn(2, 3, 256, 256, device=DEV)
    gt = torch.tensor([[[10., 12., 160., 170.], [30., 30., 220., 110.]], [[5., 5., 125., 130.], [64., 20., 200., 190.]]], device=DEV)
    labels = torch.tensor([[3, 7], [1, 20]], device=DEV)
    gen = FCOSGenTargets([8, 16, 32, 64, 128], [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]])
    crit = FCOSLoss("giou")
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}

    def step(amp):
        model.load_state_dict(sd0)
        model.zero_grad(set_to_none=True)
        seen = []
        real = ops.conv_call

        def spy(*a, **kw):
            seen.append(kw.get("precision", 0))
            return real(*a, **kw)
        ops.conv_call = spy
        try:
            scaler = torch.amp.GradScaler("cuda", enabled=amp)
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                out = model(x)
                loss = crit([out, gen([out, gt, labels])])[-1]
            scaler.scale(loss).backward()
        finally:
            ops.conv_call = real
        grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
        return float(loss.detach()), grads, seen

    T.STATS["stock_fallbacks"] = 0
    l32, g32, s32 = step(False)
    l16, g16, s16 = step(True)
    assert T.STATS["stock_fallbacks"] == 0
    assert all(p == 0 for p in s32) and sum(p == _lib.PREC_F16 for p in s16) >= 100
    assert abs(l16 - l32) < 1e-2 * abs(l32), (l16, l32)
    assert all(g is not None and bool(torch.isfinite(g).all()) for g in g16.values())
    scale = 65536.0     # GradScaler's initial scale: gradients come back scaled
    cos = {}
    for n in g32:
        a, b = g16[n].flatten().double() / scale, g32[n].flatten().double()
        if float(b.norm()) > 0:
            cos[n] = float((a @ b) / (a.norm() * b.norm() + 1e-30))
    # f16 operand rounding (2^-11 relative, every conv, both directions) through ~100 layers and ~30 batch-statistic BatchNorms: measured
    # median cosine 0.965 against the fp32 step's gradients, > 0.99 on the last layers; a wrong kernel gives ~0
    assert np.median(list(cos.values())) > 0.9, np.median(list(cos.values()))
    assert min(cos[n] for n in ("head.cls_logits.weight", "head.reg_pred.weight", "head.cnt_logits.weight")) > 0.99, cos


@pytest.mark.parametrize("case", [
    # Cin, Cout, k, stride, pad, dil, hw
    (64, 256, 1, 1, 0, 1, [(20, 24)]),
    (256, 64, 1, 1, 0, 1, [(13, 9)]),
    (256, 256, 3, 1, 1, 1, [(10, 12), (5, 6), (3, 3), (1, 2)]),     # head tower over a pyramid
    (128, 128, 3, 2, 1, 1, [(21, 13)]),
    (256, 256, 3, 1, 2, 2, [(12, 12)]),
    (512, 2048, 1, 1, 0, 1, [(5, 5)]),
    (256, 32, 3, 1, 1, 1, [(10, 12), (5, 6), (3, 3)]),              # a narrow predictor (Cout <= 32: on the f16 kernel too since round 5)
    (64, 8, 1, 1, 0, 1, [(9, 7)]),
])
def test_conv_weight_gradient_f16_operands(case):
    """fd_conv2d_bwd_weight_f32 with precision = FD_PREC_F16 (the transposing LDS reads of gfx950 feed v_mfma_f32_32x32x16_f16): dW of the
    f16-rounded (x, dy) accumulated in fp32, against torch's conv2d_weight on the rounded operands."""
    Cin, Cout, k, stride, pad, dil, hw = case
    gen = torch.Generator().manual_seed(Cin + 3 * Cout + k + stride)
    B = 2
    xs = [torch.randn(B, Cin, a, b, generator=gen) for a, b in hw]
    segs = Segs.make(B, hw)
    so = ops.conv_out_segs(segs, k, stride, pad, dil)
    dys = [torch.randn(B, Cout, a, b, generator=gen) for a, b in so.level_hw()]
    xr = ops.Rows(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(DEV))
    dr = ops.Rows(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in dys]).to(DEV))
    scale = (torch.rand(Cout, generator=gen) + 0.5)
    got = ops.conv_wgrad(xr, dr, segs, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale.to(DEV), oihw=True, precision=_lib.PREC_F16).cpu()
    ref = sum(torch.nn.grad.conv2d_weight(h(x), (Cout, Cin, k, k), h(dy), stride, pad, dil) for x, dy in zip(xs, dys)) * scale.view(-1, 1, 1, 1)
    ref32 = sum(torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, stride, pad, dil) for x, dy in zip(xs, dys)) * scale.view(-1, 1, 1, 1)
    mx = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-5 * mx + 1e-5, (float((got - ref).abs().max()), mx)
    assert float((got - ref32).abs().max()) > float((got - ref).abs().max())          # the operands really were rounded to f16
    again = ops.conv_wgrad(xr, dr, segs, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale.to(DEV), oihw=True, precision=_lib.PREC_F16).cpu()
    assert torch.equal(got, again)                                                      # ordered slab reduce: bitwise reproducible


def test_conv_f16k64_relu_from_a_channel_on():
    """act_c0 on FD_TILE_F16K64: channels below it pass through, the rest take the ReLU (a fused layer whose first outputs are linear); with act_c0 = 0 the kernel takes
    its uniform path, with act_c0 inside a lane's eight channels the per-channel selects."""
    gen = torch.Generator().manual_seed(5)
    B, Cin, Cout, hw = 2, 64, 48, [(17, 19)]
    segs = Segs.make(B, hw)
    x = h(torch.randn(segs.rows, Cin, generator=gen)).to(DEV)
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / 8
    wp = ops.pack_conv_weight_f16k64(w.to(DEV))
    lin = x @ h(w).view(Cout, Cin).t().to(DEV)
    for c0 in (0, 12, 32, 48):
        y = torch.empty(segs.rows, Cout, device=DEV, dtype=torch.float16)
        ops.conv_call(ops.Rows(x.half()), segs, wp, ops.Rows(y), Cin=Cin, Cout=Cout, k=1, act=ACT_RELU, act_c0=c0, precision=_lib.PREC_F16, tile=_lib.F16K64_TILE)()
        want = lin.clone()
        want[:, c0:] = want[:, c0:].clamp_min(0)
        np.testing.assert_allclose(y.float().cpu().numpy(), want.cpu().numpy(), atol=2e-3, rtol=2e-3, err_msg=f"act_c0 {c0}")
        assert bool((y[:, c0:] >= 0).all()) and (c0 == 0 or bool((y[:, :c0] < 0).any()))


@pytest.mark.parametrize("case", [
    # Cin, Cout, k, stride, pad, dil, hw, nsplit
    (64, 256, 1, 1, 0, 1, [(20, 24)], 0),
    (72, 40, 3, 1, 1, 1, [(10, 12), (5, 6), (3, 3), (1, 2)], 0),      # widths that end inside a lane's eight channels
    (128, 136, 3, 2, 1, 1, [(21, 13)], 3),
    (256, 64, 1, 1, 0, 1, [(33, 31)], 5),                             # ranges that are no multiple of the 64-pixel K-tile
])
@pytest.mark.parametrize("io", [(True, True), (True, False), (False, True)])
def test_conv_weight_gradient_f16_maps(case, io):
    """fd_conv_wgrad_params.io_f16: x and / or dy stored as f16 (the AMP step's activation and gradient maps).  Must equal, BIT FOR BIT, the fp32-map launch on the same
    f16-valued data (same ranges, same order); channel views with NaN neighbours (the 16-byte fetches read past a view's last channels: only rows / columns of the tile
    that are never stored may see them)."""
    Cin, Cout, k, stride, pad, dil, hw, nsplit = case
    x16, d16 = io
    gen = torch.Generator().manual_seed(Cin + Cout + k + nsplit)
    B = 2
    segs = Segs.make(B, hw)
    so = ops.conv_out_segs(segs, k, stride, pad, dil)
    x = h(torch.randn(segs.rows, Cin, generator=gen)).to(DEV)
    dy = h(torch.randn(so.rows, Cout, generator=gen)).to(DEV)
    kw = dict(Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, nsplit=nsplit, oihw=True, precision=_lib.PREC_F16)
    want = ops.conv_wgrad(ops.Rows(x), ops.Rows(dy), segs, **kw)

    def buf(t, f16, co, C_):
        b = torch.full((t.shape[0], C_ + 20), float("nan"), dtype=torch.float16 if f16 else torch.float32, device=DEV)
        b[:, co:co + C_] = t
        return b
    got = ops.conv_wgrad(ops.Rows(buf(x, x16, 8, Cin), 8, Cin), ops.Rows(buf(dy, d16, 4, Cout), 4, Cout), segs, **kw)
    assert not bool(torch.isnan(got).any())
    assert torch.equal(got, want), float((got - want).abs().max())
    xs = [x[segs.m_start[i]:segs.m_start[i + 1]].reshape(B, a, b, Cin).permute(0, 3, 1, 2).cpu() for i, (a, b) in enumerate(hw)]
    ds = [dy[so.m_start[i]:so.m_start[i + 1]].reshape(B, a, b, Cout).permute(0, 3, 1, 2).cpu() for i, (a, b) in enumerate(so.level_hw())]
    ref = sum(torch.nn.grad.conv2d_weight(a, (Cout, Cin, k, k), b, stride, pad, dil) for a, b in zip(xs, ds))
    assert float((got.cpu() - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-5
