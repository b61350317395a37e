"""GPU parity of the fused LTRB IoU / GIoU loss (forward + backward) vs golden vectors and the torch oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from pytorch_object_detection_amd.model.loss import FCOSLoss, ltrb_reg_loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("mode", ["iou", "giou"])
@pytest.mark.parametrize("P", [1, 257])
def test_ltrb_loss_golden(golden, mode, P):
    g = golden("g5_ltrb_loss")
    pred = torch.from_numpy(g[f"P{P}_pred"])[None].to(DEV).requires_grad_(True)
    tgt = torch.from_numpy(g[f"P{P}_tgt"])[None].to(DEV)
    mask = torch.ones(1, P, dtype=torch.bool, device=DEV)
    loss = ltrb_reg_loss(pred, tgt, mask, mode)           # [1] = sum / P
    np.testing.assert_allclose(float(loss[0].detach()) * P, float(g[f"P{P}_{mode}_loss"]), rtol=2e-6)
    (loss.sum() * P).backward()
    np.testing.assert_allclose(pred.grad[0].cpu().numpy(), g[f"P{P}_{mode}_grad"], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("mode", ["iou", "giou"])
def test_ltrb_loss_masked_batch_vs_oracle(mode):
    gen = torch.Generator().manual_seed(5)
    B, L = 4, 8525
    pred = torch.exp(torch.randn(B, L, 4, generator=gen)) * 8
    tgt = torch.exp(torch.randn(B, L, 4, generator=gen)) * 8
    mask = torch.rand(B, L, generator=gen) < 0.03
    mask[2] = False                                       # image with no positives: loss 0, num_pos clamps to 1
    p_ref = pred.clone().requires_grad_(True)
    fn = R.giou_loss if mode == "giou" else R.iou_loss
    ref = torch.stack([fn(p_ref[b][mask[b]], tgt[b][mask[b]]) / mask[b].sum().clamp(min=1) for b in range(B)])
    ref.mean().backward()
    p = pred.to(DEV).requires_grad_(True)
    out = ltrb_reg_loss(p, tgt.to(DEV), mask.to(DEV), mode)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-7)
    out.mean().backward()
    np.testing.assert_allclose(p.grad.cpu().numpy(), p_ref.grad.numpy(), rtol=2e-5, atol=1e-8)
    assert p.grad[2].abs().sum() == 0


def test_fcosloss_reg_term_golden(golden):
    g = golden("g67_targets_loss")
    name = "voc_his"
    reg = [torch.from_numpy(g[f"{name}_reg{i}"]).to(DEV) for i in range(5)]
    got = FCOSLoss("giou").reg_loss(reg, torch.from_numpy(g[f"{name}_reg_t"]).to(DEV), torch.from_numpy(g[f"{name}_cnt_t"]).to(DEV))
    np.testing.assert_allclose(float(got), g[f"{name}_giou_losses"][2], rtol=2e-6)


@pytest.mark.parametrize("name", ["voc_his", "voc_fcos"])
def test_gen_targets_golden(golden, name):
    from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
    g = golden("g67_targets_loss")
    outs = [[torch.from_numpy(g[f"{name}_{n}{i}"]).to(DEV) for i in range(5)] for n in ("cls", "cnt", "reg")]
    gen = FCOSGenTargets([int(s) for s in g["strides"]], g[f"{name}_ranges"].tolist())
    cls_t, cnt_t, reg_t = gen([outs, torch.from_numpy(g["gt"]).to(DEV), torch.from_numpy(g["labels"]).to(DEV)])
    np.testing.assert_array_equal(cls_t.cpu().numpy(), g[f"{name}_cls_t"])
    np.testing.assert_array_equal(reg_t.cpu().numpy(), g[f"{name}_reg_t"])
    np.testing.assert_allclose(cnt_t.cpu().numpy(), g[f"{name}_cnt_t"], rtol=1e-6)


def test_gen_targets_full_size_vs_oracle():
    gen = torch.Generator().manual_seed(9)
    B, M = 4, 8
    hw = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
    strides = [8, 16, 32, 64, 128]
    ranges = [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]]
    c = torch.rand(B, M, 2, generator=gen) * 512
    s = torch.exp(torch.rand(B, M, 2, generator=gen) * 4 + 2)
    gt = torch.round(torch.cat([c - s / 2, c + s / 2], -1).clamp(0, 511))
    labels = torch.randint(1, 21, (B, M), generator=gen)
    gt[1, 5:] = -1; labels[1, 5:] = -1
    gt[2, :] = -1; labels[2, :] = -1                       # image without any GT
    exp = R.gen_targets(hw, strides, ranges, gt, labels)
    from pytorch_object_detection_amd import ops
    got = ops.fcos_gen_targets(gt.to(DEV), labels.to(DEV), hw, strides, ranges)
    np.testing.assert_array_equal(got[0].cpu().numpy(), exp[0].numpy())
    np.testing.assert_array_equal(got[2].cpu().numpy(), exp[2].numpy())
    np.testing.assert_allclose(got[1].cpu().numpy(), exp[1].numpy(), rtol=1e-6)
    assert (got[0][2] == 0).all() and (got[1][2] == -1).all()


@pytest.mark.parametrize("name", ["voc_his", "voc_fcos"])
@pytest.mark.parametrize("mode", ["giou", "iou"])
def test_fcosloss_forward_backward_golden(golden, name, mode):
    g = golden("g67_targets_loss")
    leaves = [[torch.from_numpy(g[f"{name}_{n}{i}"]).to(DEV).requires_grad_(True) for i in range(5)] for n in ("cls", "cnt", "reg")]
    tg = [torch.from_numpy(g[f"{name}_{k}"]).to(DEV) for k in ("cls_t", "cnt_t", "reg_t")]
    res = FCOSLoss(mode)([leaves, tg])
    np.testing.assert_allclose([float(r.detach()) for r in res], g[f"{name}_{mode}_losses"], rtol=3e-6)
    res[3].backward()
    for i in range(5):
        np.testing.assert_allclose(leaves[0][i].grad.cpu().numpy(), g[f"{name}_{mode}_gcls{i}"], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(leaves[1][i].grad.cpu().numpy(), g[f"{name}_{mode}_gcnt{i}"], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(leaves[2][i].grad.cpu().numpy(), g[f"{name}_{mode}_greg{i}"], rtol=2e-5, atol=1e-8)


def test_focal_bce_full_size_vs_oracle():
    gen = torch.Generator().manual_seed(13)
    B, L, C = 2, 8525, 80
    logits = torch.randn(B, L, C, generator=gen) * 3 - 2
    logits[0, :50] = 30.0                                   # saturated sigmoid: the upper clip (== 1.0 in fp32) region
    logits[0, 50:100] = -30.0                               # below the 5e-6 lower clip: gradient is cut
    labels = (torch.rand(B, L, generator=gen) < 0.02).long() * torch.randint(1, C + 1, (B, L), generator=gen)
    from pytorch_object_detection_amd import ops
    l_ref = logits.clone().requires_grad_(True)
    ref = torch.stack([R.focal_loss(l_ref[b], (torch.arange(1, C + 1)[None] == labels[b][:, None]).float()) for b in range(B)])
    ref.sum().backward()
    got = ops.focal_loss_fwd(logits.to(DEV), labels.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), ref.detach().numpy(), rtol=2e-6)
    grad = ops.focal_loss_bwd(logits.to(DEV), labels.to(DEV), torch.ones(B, device=DEV))
    np.testing.assert_allclose(grad.cpu().numpy(), l_ref.grad.numpy(), rtol=3e-5, atol=1e-9)
