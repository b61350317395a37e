"""GPU parity of the HIP autograd nodes that complete the train step (SiLU / ReLU with saved pre-activation, max-pool + add,
nearest-upsample + add, squeeze-excitation, BatchNorm in training mode on the GroupNorm kernels) against plain PyTorch fp32
autograd of the same op, and of the rows-based HisBlock / FPN training forward against the stock-op forward."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pytorch_object_detection_amd import train_ops as T
from pytorch_object_detection_amd._lib import ACT_RELU, ACT_SILU, Segs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rows(x):           # NCHW cpu -> [B*H*W, C] cuda leaf
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV).requires_grad_(True)


def nchw(r, B, H, W):
    return r.detach().cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2)


def close(a, b, tol=2e-5):
    s = float(b.abs().max()) + 1e-12
    np.testing.assert_allclose(a.numpy() / s, b.numpy() / s, atol=tol)


@pytest.mark.parametrize("act", [ACT_SILU, ACT_RELU])
def test_act_rows_autograd_on_a_channel_slice(act):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(50, 24, generator=g)
    gy = torch.randn(50, 16, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = (F.silu if act == ACT_SILU else F.relu)(xr[:, 4:20])
    ref.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    out = T.act_rows(xd[:, 4:20], act)           # a view: read in place, no copy
    out.backward(gy.to(DEV))
    close(out.detach().cpu(), ref.detach())
    close(xd.grad.cpu(), xr.grad)


@pytest.mark.parametrize("shape", [(2, 8, 12, 10), (1, 16, 13, 21), (3, 4, 2, 2)])
def test_pool_add_and_up_add_autograd(shape):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, C, H, W, generator=g).round(decimals=1)       # rounded: windows with tied maxima exist
    Ho, Wo = H // 2, W // 2
    add = torch.randn(B, C, Ho, Wo, generator=g)
    gy = torch.randn(B, C, Ho, Wo, generator=g)
    xr, ar = x.clone().requires_grad_(True), add.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 2, 2) + ar
    ref.backward(gy)
    xd, ad = rows(x), rows(add)
    out = T._PoolAddRows.apply(xd, ad, (B, H, W, 2, 2, 0))
    out.backward(rows(gy).detach())
    close(nchw(out, B, Ho, Wo), ref.detach())
    close(nchw(xd.grad, B, H, W), xr.grad)          # incl. the tie rule: the first maximum of a window gets the gradient
    close(nchw(ad.grad, B, Ho, Wo), ar.grad)
    # upsample + add
    lat = torch.randn(B, C, 2 * H, 2 * W, generator=g)
    gy2 = torch.randn(B, C, 2 * H, 2 * W, generator=g)
    xr2, lr = x.clone().requires_grad_(True), lat.clone().requires_grad_(True)
    ref2 = F.interpolate(xr2, scale_factor=2.0, mode="nearest") + lr
    ref2.backward(gy2)
    xd2, ld = rows(x), rows(lat)
    out2 = T._UpAddRows.apply(xd2, ld, (B, H, W))
    out2.backward(rows(gy2).detach())
    close(nchw(out2, B, 2 * H, 2 * W), ref2.detach())
    close(nchw(xd2.grad, B, H, W), xr2.grad)
    close(nchw(ld.grad, B, 2 * H, 2 * W), lr.grad)


def test_maxpool3x3_stride2_backward_overlapping_windows():
    from pytorch_object_detection_amd import ops
    B, C, H, W = 2, 8, 11, 14
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g).round(decimals=1)             # many ties
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xd = rows(x).detach()
    gx = torch.empty_like(xd)
    ops.maxpool_bwd(ops.Rows(xd), ops.Rows(rows(gy).detach()), ops.Rows(gx), B, H, W, 3, 2, 1)
    close(nchw(gx, B, H, W), xr.grad)


@pytest.mark.parametrize("C,r,HW", [(128, 4, (10, 12)), (16, 4, (5, 5)), (144, 24, (3, 7))])
def test_se_rows_autograd(C, r, HW):
    from pytorch_object_detection_amd.model.modules.modules import SEBlock
    torch.manual_seed(4)
    B, (H, W) = 3, HW
    se = SEBlock(C, r) if r == 4 else None
    if se is None:
        se = SEBlock(C, 4)
        se.excitation[0] = torch.nn.Conv2d(C, C // r, 1)
        se.excitation[2] = torch.nn.Conv2d(C // r, C, 1)
    x = torch.randn(B, C, H, W)
    gy = torch.randn(B, C, H, W)
    xr = x.clone().requires_grad_(True)
    ref = xr * se.excitation(xr.mean((2, 3), keepdim=True))
    ref.backward(gy)
    gref = {n: p.grad.clone() for n, p in se.named_parameters()}
    se.zero_grad()
    se.to(DEV)
    xd = rows(x)
    out = T.se_rows(se, xd, B, H * W)
    out.backward(rows(gy).detach())
    close(nchw(out, B, H, W), ref.detach())
    close(nchw(xd.grad, B, H, W), xr.grad)
    for n, p in se.named_parameters():
        assert p.grad.shape == gref[n].shape
        close(p.grad.cpu(), gref[n], 5e-5)


@pytest.mark.parametrize("C,act,shape", [(128, ACT_SILU, (4, 9, 7)), (256, ACT_RELU, (2, 16, 16)), (16, ACT_RELU, (16, 2, 2))])
def test_batchnorm_train_rows_matches_nn_batchnorm(C, act, shape):
    B, H, W = shape
    torch.manual_seed(5)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5); bn.bias.copy_(torch.randn(C) * 0.1)
        bn.running_mean.copy_(torch.randn(C) * 0.1); bn.running_var.copy_(torch.rand(C) + 0.5)
    x = torch.randn(B, C, H, W) * 2 + 0.3
    gy = torch.randn(B, C, H, W)
    ref_bn = torch.nn.BatchNorm2d(C)
    ref_bn.load_state_dict(bn.state_dict())
    ref_bn.train()
    xr = x.clone().requires_grad_(True)
    ref = (F.silu if act == ACT_SILU else F.relu)(ref_bn(xr))
    ref.backward(gy)
    bn.to(DEV).train()
    assert T._bn_train_ok(bn, torch.empty(1, device=DEV))
    xd = rows(x)
    out = T.batchnorm_train_rows(bn, xd, act)
    out.backward(rows(gy).detach())
    close(nchw(out, B, H, W), ref.detach())
    close(nchw(xd.grad, B, H, W), xr.grad, 1e-4)
    close(bn.weight.grad.cpu(), ref_bn.weight.grad, 1e-4)
    close(bn.bias.grad.cpu(), ref_bn.bias.grad, 1e-4)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref_bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref_bn.running_var.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bn_mode", ["frozen", "training"])
def test_hisblock_and_fpn_rows_training_forward_match_the_stock_ops(bn_mode):
    """The all-HIP rows path of HalfInvertedStageFPN.train_forward against the stock-op path (train_ops._STOCK): same outputs,
    same input / parameter gradients, same running statistics, in both BatchNorm modes."""
    from pytorch_object_detection_amd.model.od.HISFcos import HalfInvertedStageFPN
    torch.manual_seed(6)
    fpn = HalfInvertedStageFPN([64, 128, 256], 64).to(DEV)
    for p in fpn.parameters():
        if p.dim() == 4:
            torch.nn.init.normal_(p, std=(2.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
    for m in fpn.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.train(bn_mode == "training")
            if bn_mode == "frozen":
                for p in m.parameters():
                    p.requires_grad = False
    state0 = {k: v.clone() for k, v in fpn.state_dict().items()}
    feats = [torch.randn(4, c, s, s, device=DEV).to(memory_format=torch.channels_last) for c, s in ((64, 32), (128, 16), (256, 8))]
    res = []
    for stock in (False, True):
        fpn.load_state_dict(state0)
        fpn.zero_grad()
        xs = [f.clone().requires_grad_(True) for f in feats]
        T._STOCK = stock
        try:
            if stock:
                assert fpn.train_forward_rows(xs) is None
                out = fpn.train_forward(xs)
            else:
                out = fpn.train_forward_rows(xs)          # (called once: a training-mode BatchNorm updates its running statistics)
                assert out is not None
            torch.manual_seed(7)
            sum((t * torch.randn(t.shape, device=DEV)).sum() for t in out).backward()
        finally:
            T._STOCK = False
        res.append(([t.detach().clone() for t in out], [x.grad.clone() for x in xs],
                    {n: p.grad.clone() for n, p in fpn.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in fpn.state_dict().items() if "running" in k}))
    (o1, g1, p1, r1), (o2, g2, p2, r2) = res
    for a, b in zip(o1, o2):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=3e-4, rtol=3e-4)
    for a, b in zip(g1, g2):
        close(a.cpu(), b.cpu(), 1e-3)
    assert p1.keys() == p2.keys() and len(p1) > 30
    gmax = max(float(v.abs().max()) for v in p2.values())
    for n in p1:
        if float(p2[n].abs().max()) < 1e-4 * gmax:       # a conv bias in front of a batch-statistic BatchNorm: zero gradient but for rounding
            assert float(p1[n].abs().max()) < 1e-3 * gmax, n
            continue
        close(p1[n].cpu(), p2[n].cpu(), 2e-3)
    for k in r1:
        np.testing.assert_allclose(r1[k].cpu().numpy(), r2[k].cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
