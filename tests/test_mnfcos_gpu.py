"""MNFCOS (reference model/od/MNFcos.py, the detector config/main.yaml:2 selects) on the HIP path: the blocks
against the reference-generated fixture g10 where the reference runs as shipped (MNHeadFCOS, MNBlock k = 3), against the oracle's
repaired restatement elsewhere (k = 5 / 7 blocks, the light-weight FPN, the whole model)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from pytorch_object_detection_amd._lib import FdError
from pytorch_object_detection_amd.bulider import Builder, load_config
from pytorch_object_detection_amd.model.modules.head import ClipBoxes, FCOSHead
from pytorch_object_detection_amd.model.modules.modules import MNBlock
from pytorch_object_detection_amd.model.od import MNFCOS
from pytorch_object_detection_amd.model.od.MNFcos import LieghtWeightFeaturePyramid_old, MNHeadFCOS
from test_model_gpu import assert_same_detections, randomize_norms

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(atol=1e-4, rtol=1e-4)


def test_mn_head_and_k3_blocks_vs_reference_fixture(golden):
    g = golden("g10_mnfcos_parts")
    head = MNHeadFCOS(32, 20, 0.01).eval()
    sd = {k[len("sd.head."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.head.")}
    assert set(sd) == set(head.state_dict())                       # the reference's parameter names
    head.load_state_dict(sd, strict=True)
    head.to(DEV)
    with torch.no_grad():
        cls, cnt, reg = head([torch.from_numpy(g[f"f{i}"]).to(DEV) for i in range(5)])
    for i in range(5):
        np.testing.assert_allclose(cls[i].cpu().numpy(), g[f"cls{i}"], err_msg=f"cls{i}", **TOL)
        np.testing.assert_allclose(cnt[i].cpu().numpy(), g[f"cnt{i}"], err_msg=f"cnt{i}", **TOL)
        np.testing.assert_allclose(reg[i].cpu().numpy(), g[f"reg{i}"], err_msg=f"reg{i}", **TOL)
    for name, k, d in (("mnb_k3d1", 3, 1), ("mnb_k3d2", 3, 2)):
        blk = MNBlock(32, 32, k, d, 2).eval()
        blk.load_state_dict({kk[len(name) + 4:]: torch.from_numpy(g[kk]) for kk in g.files if kk.startswith(name + ".sd.")}, strict=True)
        blk.to(DEV)
        with torch.no_grad():
            y = blk(torch.from_numpy(g[name + ".x"]).to(DEV))
        np.testing.assert_allclose(y.cpu().numpy(), g[name + ".y"], err_msg=name, **TOL)


@pytest.mark.parametrize("k,d", [(3, 1), (3, 2), (5, 1), (5, 2), (7, 1)])
def test_mn_block_vs_oracle(k, d):
    """every (kernel, dilation) MNFCOS uses, odd map sizes, C = 128 (alpha = 2): HIP launches vs the oracle's repaired restatement"""
    torch.manual_seed(k * 10 + d)
    blk = MNBlock(128, 128, k, d, 2).eval()
    randomize_norms(blk, k + d)
    sd = {kk: v.clone() for kk, v in blk.state_dict().items()}
    x = torch.randn(2, 128, 11, 7)
    with torch.no_grad():
        ref = R.mn_block(sd, "", x, k, d)
        got = blk.to(DEV)(x.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), **TOL)
    blk.train()
    with pytest.raises(FdError, match="forward only"):
        blk(x.to(DEV))


def test_mn_fpn_vs_oracle():
    torch.manual_seed(4)
    fpn = LieghtWeightFeaturePyramid_old([128, 64, 32], 32).eval()
    randomize_norms(fpn, 5)
    sd = {"FeaturePyramidNetwork." + k: v.clone() for k, v in fpn.state_dict().items()}
    feats = [torch.randn(2, 32, 16, 16), torch.randn(2, 64, 8, 8), torch.randn(2, 128, 4, 4)]
    with torch.no_grad():
        ref = R.mn_fpn(sd, feats)
        got = fpn.to(DEV)([f.to(DEV) for f in feats])
    assert [tuple(t.shape[2:]) for t in got] == [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    for i in range(5):
        np.testing.assert_allclose(got[i].cpu().numpy(), ref[i].numpy(), err_msg=f"P{i + 3}", **TOL)


def test_full_mnfcos_vs_oracle_and_builder(tmp_path):
    """Builder(load_config(...)) with model = MNFCOS (what the reference's config/main.yaml selects) -> 2 x 3 x 128 x 128 -> every head
    output within 1e-4 of the oracle, detections = the oracle post-process of the device's outputs; train() + forward raises."""
    import os
    import pytorch_object_detection_amd as pkg
    cfg_dir = os.path.join(os.path.dirname(pkg.__file__), "config")
    main = tmp_path / "config" / "main.yaml"
    main.parent.mkdir()
    main.write_text(open(os.path.join(cfg_dir, "main.yaml")).read().replace("dataset: COCO", "dataset: VOC").replace("model: HISFCOS", "model: MNFCOS")
                    .replace("config/voc.yaml", os.path.join(cfg_dir, "voc.yaml")))
    cfg = load_config(str(main))
    assert cfg["model"]["name"] == "MNFCOS"
    torch.manual_seed(6)
    model = Builder(cfg).model_build().eval()
    assert isinstance(model, MNFCOS) and model.head.cls_logits.weight.shape == (cfg["dataset_setting"]["class_num"], 256, 1, 1)
    randomize_norms(model, 7)
    with torch.no_grad():                       # the default init drives the predictors to their bias: make the towers matter
        for m in (model.head.cls_logits, model.head.cnt_logits, model.head.reg_pred):
            m.weight.mul_(3.0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randn(2, 3, 128, 128)
    with torch.no_grad():
        ref = R.mnfcos_forward(sd, x)
    model.to(DEV)
    xd = x.to(DEV)
    out = model(xd)
    for name, o, r in zip(("cls", "cnt", "reg"), out, ref):
        assert len(o) == 5
        for i in range(5):
            assert tuple(o[i].shape) == tuple(r[i].shape)
            np.testing.assert_allclose(o[i].cpu().numpy(), r[i].numpy(), err_msg=f"{name}{i}", **TOL)
    strides = cfg["MNFCOS"]["stride"]
    s, c, b, counts = FCOSHead(0.05, 0.6, 1000, strides).detect_padded(out)
    b = ClipBoxes()(xd, b)
    exp = R.fcos_detect([[t.cpu() for t in grp] for grp in out], strides, 0.05, 0.6, 1000, (128, 128))
    for bi in range(2):
        n = int(counts[bi])
        assert n == len(exp[bi][0]) and n > 0
        assert_same_detections(s[bi, :n].cpu().numpy(), c[bi, :n].cpu().numpy(), b[bi, :n].cpu().numpy(), *exp[bi])
    model.train()
    with pytest.raises(FdError, match="inference-only"):
        model(xd)
