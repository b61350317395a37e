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
    np.testing.assert_allclose(float(loss[0]) * P, float(g[f"P{P}_{mode}_loss"]), rtol=2e-6)
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
