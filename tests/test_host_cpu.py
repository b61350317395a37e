"""CPU-side checks: the C-ABI library loads and exports every symbol of include/fcosdet.h, the ctypes structs match
the C layout, host logic (config, builder, state_dict contract, weight packing, BN folding) and the no-fallback rule."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from pytorch_object_detection_amd import _lib, ops  # noqa: E402
from pytorch_object_detection_amd.bulider import Builder, load_config  # noqa: E402
from pytorch_object_detection_amd.model.modules.head import ClipBoxes, FCOSGenTargets, FCOSHead  # noqa: E402
from pytorch_object_detection_amd.model.od import FCOS, HalfInvertedStageFCOS  # noqa: E402


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fcosdet.h")).read()
    declared = set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/fcosdet.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.fd_version() >= 100


def test_struct_layout_matches_c():
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "fcosdet.h"
    int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(fd_segs), sizeof(fd_conv_params),
        offsetof(fd_conv_params, x_cs), offsetof(fd_conv_params, seg_param), offsetof(fd_conv_params, in),
        offsetof(fd_conv_params, tile), offsetof(fd_conv_params, ksplit), offsetof(fd_conv_params, workspace),
        offsetof(fd_conv_params, workspace_bytes), offsetof(fd_conv_params, precision));
      printf("%zu %zu %zu %zu %zu\n", offsetof(fd_conv_params, x2), offsetof(fd_conv_params, x2_W), sizeof(fd_b2b_params), offsetof(fd_b2b_params, x_cs),
        offsetof(fd_b2b_params, rows));
      printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(fd_conv_wgrad_params), offsetof(fd_conv_wgrad_params, dw),
        offsetof(fd_conv_wgrad_params, workspace), offsetof(fd_conv_wgrad_params, nsplit), offsetof(fd_conv_wgrad_params, layout),
        offsetof(fd_conv_wgrad_params, scale), offsetof(fd_conv_wgrad_params, in)); return 0; }'''
    exe = os.path.join(ROOT, "oracle", "_build", "abi_probe")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    vals = [int(v) for v in subprocess.check_output([exe]).split()]
    P = _lib.ConvParams
    Q = _lib.WgradParams
    B = _lib.B2BParams
    assert vals == [ctypes.sizeof(_lib.Segs), ctypes.sizeof(P), P.x_cs.offset, P.seg_param.offset, P.segs.offset,
                    P.tile.offset, P.ksplit.offset, P.workspace.offset, P.workspace_bytes.offset, P.precision.offset,
                    P.x2.offset, P.x2_W.offset, ctypes.sizeof(B), B.x_cs.offset, B.rows.offset,
                    ctypes.sizeof(Q), Q.dw.offset, Q.workspace.offset, Q.nsplit.offset, Q.layout.offset, Q.scale.offset,
                    Q.segs.offset]


def test_segs_table():
    s = _lib.Segs.make(2, [(4, 4), (2, 2), (1, 1)])
    assert s.nseg == 3 and s.rows == 2 * (16 + 4 + 1)
    assert list(s.m_start)[:4] == [0, 32, 40, 42]
    assert s.level_hw() == [(4, 4), (2, 2), (1, 1)]


def test_config_and_builder():
    cfg = load_config()
    assert cfg["model"]["name"] == "HISFCOS" and cfg["dataset_setting"]["class_num"] == 80
    assert cfg["HISFCOS"]["CannelofBackbone"] == [512, 1024, 2048]
    assert cfg["HISFCOS"]["range"][0] == [-1, 32] and cfg["FCOS"]["range"][0] == [-1, 64]
    m = Builder(cfg).model_build()
    assert isinstance(m, HalfInvertedStageFCOS)
    opt = Builder(cfg).opt_build(m)
    assert isinstance(opt, torch.optim.SGD) and opt.defaults["weight_decay"] == 1e-4
    cfg["model"]["name"] = "SSD300"
    with pytest.raises(NotImplementedError):
        Builder(cfg).model_build()


def test_state_dict_contract():
    m = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256)
    sd = m.state_dict()
    # reference parameter counts (HISFcos.py:247-248) and both backbone key sets
    assert sum(p.numel() for p in m.fpn.parameters()) == 7648224
    assert sum(p.numel() for p in m.head.parameters()) == 1507358
    for k in ("backbone.conv1.weight", "backbone.layer1.0.conv1.weight", "backbone.extract_feature.layer4.2.conv3.weight",
              "fpn.tf1.weight", "fpn.gn3.running_mean", "fpn.HisBlock7.conv1_2.excitation.2.bias", "head.pw2.bias",
              "head.cls_conv.1.weight", "head.scale_exp.4.scale"):
        assert k in sd, k
    assert sd["head.cls_logits.bias"][0].item() == pytest.approx(-np.log(99.0))
    assert not any(p.requires_grad for n, p in m.named_parameters() if ".bn" in n or ".gn" in n and "fpn" in n)
    # DDP checkpoints carry a 'module.' prefix that the reference strips (test.py:273-281)
    m.load_state_dict({k: v for k, v in sd.items()})
    f = FCOS([2048, 1024, 512], 20, 256)
    assert "FPN.P6_c1.bias" in f.state_dict() and "head.cls_branch.9.weight" in f.state_dict()


def test_mnfcos_containers_and_builder():
    """MNFCOS (the detector the reference's config/main.yaml selects): the reference's state_dict names (pinned by fixture g10 for the
    head), the 'same' padding repair of the k = 5 / 7 MNBlocks, the Builder entry."""
    from pytorch_object_detection_amd.model.od import MNFCOS
    from pytorch_object_detection_amd.model.modules.modules import MNBlock
    m = MNFCOS([2048, 1024, 512], 20, 256)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_mnfcos_parts.npz"))
    ref_head = {k[len("sd.head."):]: g[k].shape for k in g.files if k.startswith("sd.head.")}
    assert set(ref_head) == set(m.head.state_dict())
    for k in ("FeaturePyramidNetwork.C5PW.bias", "FeaturePyramidNetwork.MNB7.DilatedDepthWiseConv.weight", "FeaturePyramidNetwork.MNB1_P3.PW2.weight",
              "head.block2.BN.running_var", "backbone.extract_feature.layer4.2.conv3.weight"):
        assert k in m.state_dict(), k
    assert m.FeaturePyramidNetwork.MNB7.DilatedDepthWiseConv.weight.shape == (256, 1, 7, 7)
    for k, d in ((3, 1), (3, 2), (5, 1), (5, 2), (7, 1)):
        assert MNBlock(8, 8, k, d, 2).DilatedDepthWiseConv.padding == (d * (k - 1) // 2,) * 2
    cfg = load_config()
    cfg["model"]["name"] = "MNFCOS"
    assert isinstance(Builder(cfg).model_build(), MNFCOS) and cfg["MNFCOS"]["CannelofBackbone"] == [2048, 1024, 512]


def test_winograd_tile_count_and_kernel_choice():
    """Host logic of the Winograd path: the tile enumeration the kernel walks (tools/wino_emul.py proves the same formula against
    F.conv2d) and the per-layer choice between the Winograd and the direct kernel (ops.wino_choice)."""
    from pytorch_object_detection_amd import ops
    seg = _lib.Segs.make(2, [(5, 5), (3, 4), (1, 1)])
    assert ops.wino_tiles(seg, 1) == 2 * (9 + 4 + 1)                 # ceil(H/2) x ceil(W/2) per image and level
    assert ops.wino_tiles(_lib.Segs.make(1, [(8, 7)]), 2) == 4 * 2 * 2  # dilation 2: four parity classes of a 4 x 4 sub-lattice
    assert ops.wino_ok(256, 512, 3, 1, 1, 1) and ops.wino_ok(256, 256, 3, 1, 2, 2) and ops.wino_ok(256, 8, 3, 1, 1, 1)
    assert not ops.wino_ok(256, 256, 3, 2, 1, 1) and not ops.wino_ok(256, 256, 1, 1, 0, 1) and not ops.wino_ok(12, 32, 3, 1, 1, 1)
    pyr = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
    assert ops.wino_choice(_lib.Segs.make(16, pyr), 256, 512, 1) == (True, 1)         # the head tower at the bench shape
    assert ops.wino_choice(_lib.Segs.make(16, [(160, 160)]), 64, 64, 1)[0]            # layer1 conv2
    use, ks = ops.wino_choice(_lib.Segs.make(1, [(32, 32)]), 256, 256, 1)             # batch-1 layer3 conv2: few tiles -> split-K or direct
    assert (not use) or ks > 1
    use, ks = ops.wino_choice(_lib.Segs.make(16, [(5, 5)]), 256, 256, 2)              # the 5 x 5 pyramid level: 32 workgroups
    assert (not use) or ks > 1
    assert not ops.wino_choice(_lib.Segs.make(16, [(5, 5)]), 256, 256, 2, allow_split=False)[0]
    assert ops.wino_preferred(_lib.Segs.make(16, pyr), 256, 512, 1)


def test_weight_packing_and_bn_fold():
    w = torch.randn(8, 32, 3, 3)
    p = ops.pack_conv_weight(w)
    assert p.shape == (8, 1, 3, 3, 32) and torch.equal(p[3, 0, 1, 2], w[3, :, 1, 2])
    w64 = torch.randn(4, 64, 3, 3)
    assert torch.equal(ops.pack_conv_weight(w64)[2, 1, 0, 2], w64[2, 32:, 0, 2])
    st = ops.pack_stem_weight(torch.randn(64, 3, 7, 7))
    assert st.shape == (64, 7, 8, 4) and st[:, :, 7].abs().sum() == 0 and st[..., 3].abs().sum() == 0
    d = ops.pack_dw_weight(torch.arange(36.).reshape(4, 1, 3, 3))
    assert d.shape == (9, 4) and d[5, 2] == 2 * 9 + 5
    g, b, mu, var, cb = torch.rand(6) + .5, torch.randn(6), torch.randn(6), torch.rand(6) + .5, torch.randn(6)
    sc, sf = ops.fold_bn(g, b, mu, var, 1e-5, cb)
    z = torch.randn(10, 6)
    np.testing.assert_allclose((z * sc + sf).numpy(), ((z + cb - mu) / torch.sqrt(var + 1e-5) * g + b).numpy(), rtol=1e-5, atol=1e-6)


def test_no_cpu_fallback():
    m = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).eval()
    with pytest.raises(_lib.FdError, match="GPU only"):
        m(torch.zeros(1, 3, 64, 64))
    head = FCOSHead(0.05, 0.6, 1000, [8, 16, 32, 64, 128])
    outs = [[torch.zeros(1, c, 4, 4)] for c in (20, 1, 4)]
    with pytest.raises(_lib.FdError, match="GPU only"):
        head(outs)
    with pytest.raises(_lib.FdError):
        ops.clip_boxes_(torch.zeros(1, 4, 4), 10, 10)
    with pytest.raises(_lib.FdError, match="GPU only"):
        FCOSGenTargets([8], [[-1, 64]])([[[torch.zeros(1, 20, 4, 4)]], torch.zeros(1, 2, 4), torch.zeros(1, 2, dtype=torch.long)])


def test_abi_argument_errors_without_gpu():
    lib = _lib.lib()
    # invalid arguments are rejected on the host before any launch
    assert lib.fd_clip_boxes(None, 4, 10, 10, None) == _lib.lib().fd_clip_boxes(None, 4, 10, 10, None) < 0
    assert b"null" in lib.fd_last_error()
    assert lib.fd_fcos_topk(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 10, 2000, ctypes.c_void_p(16),
                            ctypes.c_void_p(16), ctypes.c_void_p(16), None, None, None) < 0
    p = _lib.ConvParams()
    assert lib.fd_conv2d_nhwc_f32(ctypes.byref(p), None) == -1
    assert lib.fd_groupnorm_workspace_bytes(ctypes.byref(_lib.Segs.make(2, [(8, 8), (4, 4)])), 32) == 2 * 2 * (256 + 1) * 32 * 16   # chunk partials + (mean, rstd)


def test_winograd4_grid_queries_without_gpu():
    """fd_conv_workgroups / fd_conv_workgroups_live are host arithmetic: the HISFCOS head tower at 16 x 640 x 640 (256 -> 512 over 80 / 40 / 20 / 10 / 5) is
    8 608 tiles = 269 M tiles x 8 cout tiles = 2 152 workgroups in a grid of 2 176 (eight XCD shares of 34 M tiles); its first 2 048 workgroups -- 8 whole
    rounds on 256 CUs, what engine.TOWER_TAIL_SPLIT launches as the main launch -- hold 2 040 of them."""
    lib = _lib.lib()
    p = _lib.ConvParams()
    p.segs = _lib.Segs.make(16, [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)])
    p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad, p.dil, p.tile = 256, 512, 3, 3, 1, 1, 1, _lib.WINO4_TILE
    assert lib.fd_conv_workgroups(ctypes.byref(p)) == 2176 and lib.fd_conv_workgroups_live(ctypes.byref(p)) == 2152
    p.wg_first, p.wg_count = 0, 2048
    assert lib.fd_conv_workgroups_live(ctypes.byref(p)) == 2040
    p.wg_first, p.wg_count = 2048, 128
    assert lib.fd_conv_workgroups_live(ctypes.byref(p)) == 112
    p.wg_first, p.wg_count = 2048, 136
    assert lib.fd_conv_workgroups_live(ctypes.byref(p)) < 0 and b"slice" in lib.fd_last_error()
    p.wg_first, p.wg_count, p.dil, p.pad = 0, 0, 2, 2                 # dilation 2: four parity classes per image, tiles per class rounded up
    assert lib.fd_conv_workgroups_live(ctypes.byref(p)) == -(-16 * 4 * (100 + 25 + 9 + 4 + 1) // 32) * 8
    p.tile = 4
    assert lib.fd_conv_workgroups(ctypes.byref(p)) < 0                  # only the F(4x4) kernel's grid can be sliced


def test_backward_abi_argument_errors_without_gpu():
    """The train-step entry points validate on the host too (no launch happens for bad arguments)."""
    lib = _lib.lib()
    p = _lib.WgradParams()
    assert lib.fd_conv2d_bwd_weight_f32(ctypes.byref(p), None) < 0
    assert lib.fd_conv_wgrad_workspace_bytes(4096, 256, 256, 3, 3) >= 256 * 256 * 9 * 4
    assert lib.fd_conv_wgrad_workspace_bytes(0, 256, 256, 3, 3) == -1
    assert lib.fd_pack_conv_weight_f32(None, None, None, 64, 64, 3, 3, 0, None) < 0
    assert lib.fd_pack_conv_weight_f32(ctypes.c_void_p(16), None, ctypes.c_void_p(16), 64, 48, 3, 3, 0, None) < 0   # Cin % 32
    segs = _lib.Segs.make(2, [(8, 8), (4, 4)])
    assert lib.fd_groupnorm_bwd_workspace_bytes(ctypes.byref(segs), 64) == (4 * 256 + 4) * 2 * 64 * 8
    assert lib.fd_dwconv3x3_wgrad_workspace_bytes(ctypes.byref(segs), 128) > 0
    assert lib.fd_groupnorm_act_bwd_nhwc(None, 0, 0, None, 0, 0, None, None, None, 0, 0, None, None, 64, 32, 1e-5, 0,
                                         ctypes.byref(segs), None, None, None) < 0


def test_efficientnet_containers_match_published_architecture():
    """The EfficientNet trunk is third-party arithmetic (efficientnet_pytorch 0.7.1, absent here): pin the restated
    architecture table to what IS published -- the parameter counts of B0..B7 -- and to the reference's own notes (B3
    endpoints 24/32/48/136/384), for the product containers AND the oracle's independent table."""
    from oracle import effnet_ref as E
    from pytorch_object_detection_amd.model.backbone.efficientnetv1 import EfficientNetV1, static_same_padding
    published = {0: 5288548, 1: 7794184, 2: 9109994, 3: 12233232, 4: 19341616, 5: 30389784, 6: 43040704, 7: 66347960}
    for n in (0, 3, 7):
        net = EfficientNetV1(n)
        assert sum(p.numel() for p in net.parameters()) == published[n], n
        stem, blocks, head, res = E.block_table(E.VALID_MODELS[n])
        assert len(blocks) == len(net.model._blocks) and res == net.model.image_size
        for (k, s, e, ci, co, nominal), blk in zip(blocks, net.model._blocks):
            assert (k, s, e, ci, co) == (blk.kernel, blk.stride, blk.expand, blk.cin, blk.cout)
            assert E.same_pad(nominal, k, s) == blk.pad
    b3 = EfficientNetV1(3)
    assert b3.endpoint_channels == [24, 32, 48, 136, 384]
    # static "SAME" padding computed for the nominal 300x300 input, whatever arrives (Conv2dStaticSamePadding)
    assert b3.model.stem_pad == (0, 1)
    assert [(b.kernel, b.pad) for b in b3.model._blocks if b.stride == 2] == [(3, (0, 1)), (5, (2, 2)), (3, (0, 1)), (5, (2, 2))]
    assert static_same_padding(224, 5, 2) == (1, 2) and static_same_padding(19, 5, 1) == (2, 2)
    sd = b3.state_dict()
    for k in ("model._conv_stem.weight", "model._bn0.running_var", "model._blocks.0._depthwise_conv.weight", "model._blocks.0._se_reduce.bias",
              "model._blocks.2._expand_conv.weight", "model._blocks.25._project_conv.weight", "model._blocks.25._bn2.num_batches_tracked",
              "model._conv_head.weight", "model._bn1.weight", "model._fc.bias"):
        assert k in sd, k
    assert "model._blocks.0._expand_conv.weight" not in sd          # expand ratio 1: no expansion conv
    assert sd["model._blocks.2._se_reduce.weight"].shape == (6, 144, 1, 1) and b3.model._bn0.eps == 1e-3
    f = FCOS([384, 136, 48], 80, 256, efficientnet=True, backbone_number=3)
    assert "backbone.model._blocks.7._bn1.running_mean" in f.state_dict()
    with pytest.raises(_lib.FdError, match="in_channel"):
        FCOS([2048, 1024, 512], 80, 256, efficientnet=True, backbone_number=3)
    with pytest.raises(_lib.FdError, match="GPU only"):
        b3.eval()(torch.zeros(1, 3, 64, 64))


def test_effnet_oracle_runs_and_is_deterministic():
    from oracle import effnet_ref as E
    from effnet_init import init_effnet
    from pytorch_object_detection_amd.model.backbone.efficientnetv1 import EfficientNetV1
    net = EfficientNetV1(0).eval()
    init_effnet(net, 1)
    sd = {"backbone." + k: v for k, v in net.state_dict().items()}
    x = torch.randn(1, 3, 96, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        e = E.extract_endpoints(sd, x, "efficientnet-b0")
    assert [tuple(e[f"reduction_{i}"].shape[1:]) for i in range(1, 7)] == [(16, 48, 32), (24, 24, 16), (40, 12, 8), (112, 6, 4), (320, 3, 2),
                                                                          (1280, 3, 2)]
    assert all(0.05 < float(e[f"reduction_{i}"].std()) < 5 for i in range(1, 6))       # the test init keeps every endpoint O(1)


def test_reference_checkpoint_layouts_load():
    """test.py:273-281 / Test_coco.py:206-214: checkpoints saved from a DistributedDataParallel model carry a 7-character
    'module.' prefix the scripts strip; ResNet50v2 checkpoints hold BOTH key sets (backbone.conv1 / bn1 / layer1.* and
    backbone.extract_feature.*, shared tensors in the reference's fx GraphModule).  Both layouts load strictly."""
    m = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256)
    sd = m.state_dict()
    assert "backbone.conv1.weight" in sd and "backbone.extract_feature.conv1.weight" in sd
    assert sd["backbone.layer1.0.conv1.weight"].data_ptr() == sd["backbone.extract_feature.layer1.0.conv1.weight"].data_ptr()
    ddp_ckpt = {"module." + k: v.clone() + (0.5 if v.is_floating_point() else 0) for k, v in sd.items()}
    stripped = {k[7:]: v for k, v in ddp_ckpt.items()}                      # the reference's `k[7:]`
    m2 = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256)
    missing, unexpected = m2.load_state_dict(stripped, strict=True)
    assert not missing and not unexpected
    assert torch.equal(m2.state_dict()["fpn.tf1.weight"], sd["fpn.tf1.weight"] + 0.5)
    # a checkpoint that carries only ONE of the two backbone key sets still fills the shared tensors
    one = {k: v for k, v in stripped.items() if not k.startswith(("backbone.conv1", "backbone.bn1", "backbone.layer1"))}
    m3 = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256)
    res = m3.load_state_dict(one, strict=False)
    assert not res.unexpected_keys and all(k.startswith(("backbone.conv1", "backbone.bn1", "backbone.layer1")) for k in res.missing_keys)
    assert torch.equal(m3.backbone.conv1.weight, stripped["backbone.extract_feature.conv1.weight"])
    f = FCOS([2048, 1024, 512], 20, 256)
    f.load_state_dict({k[7:]: v for k, v in {"module." + k: v for k, v in f.state_dict().items()}.items()}, strict=True)


def test_tile_table_wave_entry_is_demoted_without_fragment_packed_weights(monkeypatch):
    """the committed tile table holds an FD_TILE_WAVE64 entry whose key does not say whether the caller packed its weights in MFMA
    fragment order (fd_conv_params.w_frag).  train_ops._conv_launch never does and plans built with FD_WAVE_TILE=0 do not either: the table
    entry -- also through the nearest-batch transfer -- must then come out as the heuristic tile, not as a launch the library refuses."""
    key = "B16|208x336|32>192|k1s1p0d1|res0|xcs32|ycs192"
    monkeypatch.setattr(ops, "_TUNE_MODE", "0")
    monkeypatch.setattr(ops, "_TUNE_CACHE", {key: _lib.WAVE_TILE})
    monkeypatch.setattr(ops, "_TUNE_LOADED", True)

    class Run:
        params = _lib.ConvParams()
    p = Run.params
    for k in (key, key.replace("B16|", "B8|"), key.replace("B16|", "B32|")):          # exact entry and the 0.5x .. 2x batch transfer
        p.w_frag = None
        assert ops.autotune_conv(Run, k, 16 * 208 * 336, 192, 1) == 0 and p.tile == 0 and p.ksplit == 1
        p.w_frag = 4096                                                                # a (fake) fragment-packed weight pointer: the entry stands
        assert ops.autotune_conv(Run, k, 16 * 208 * 336, 192, 1) == _lib.WAVE_TILE and p.tile == _lib.WAVE_TILE
    # split-K bits are dropped, and reported as dropped, when the launch has no workspace or carries a row-statistics epilogue
    monkeypatch.setattr(ops, "_TUNE_CACHE", {"k": 8 | (4 << 8)})
    p.w_frag, p.workspace, p.gn_stats = None, None, None
    assert ops.autotune_conv(Run, "k", 1000, 64, 8) == 8 and p.ksplit == 1
    p.workspace, p.gn_stats = 4096, 8192
    assert ops.autotune_conv(Run, "k", 1000, 64, 8) == 8 and p.ksplit == 1
    p.gn_stats = None
    assert ops.autotune_conv(Run, "k", 1000, 64, 8) == (8 | (4 << 8)) and p.ksplit == 4


def test_fderror_carries_the_numeric_return_code():
    with pytest.raises(_lib.FdError) as ei:
        _lib.check(_lib.E_UNSUPPORTED, "probe")
    assert ei.value.rc == _lib.E_UNSUPPORTED == -2 and "(-2)" in str(ei.value)
    assert _lib.FdError("host-side").rc is None


def test_bench_roofline_object_describes_the_marked_launch():
    """bench.tower_roofline reads tile / FLOPs / Winograd divisor from the plan step the 'head.tower3x3' mark actually covers (FCOS names
    its fused tower launch head.tower0: looking up a fixed name found nothing, took tile 0 = "direct" and printed frac 2.2 for an F(4x4) launch), reports the
    whole layer beside a split grid's main launch, and attaches PMC traffic only to the workload the PMC passes were made for."""
    import types
    import bench
    rows, Cin, Cout = 16 * 8525, 256, 512
    alg = 2 * rows * Cout * Cin * 9

    def plan(name, tile, div, share=1.0, tail=False, sk=0):
        p = types.SimpleNamespace()
        p.names = ["x.pre", name] + ([name + ".tail"] if tail else []) + ["x.post"]
        p.marks = {"head.tower3x3": (1, 2)}
        p.tiles = {name: tile}
        p.step_flops = {1: int(alg * share)}
        p.step_info = {1: {"Cin": Cin, "Cout": Cout, "mfma_div": div, "family": "winograd3x3"}}
        p.segs = types.SimpleNamespace(nseg=5, rows=rows)
        if tail:
            p.step_flops[2] = alg - int(alg * share)
            p.step_info[2] = dict(p.step_info[1])
            p.tail_of = {name: {"workgroups": 2176, "main": 2048, "live": 2152, "main_share": share}}
        if sk:
            p.sk_of = {name: sk}
        return p

    wl = {"model": "FCOS", "batch": 16, "height": 640, "width": 640}
    ms = 0.9
    r = bench.tower_roofline(plan("head.tower0", 16, 4.0), ms, wl, layer_ms=ms)
    assert r["flops_per_launch"] == alg // 4 and r["algorithmic_flops_per_launch"] == alg
    assert abs(r["frac"] - alg / 4 / (ms * 1e-3) / 1e12 / bench.PEAK_F32_MFMA_TFLOPS) < 1e-3 and r["frac"] < 1.0
    assert "F(4x4" in r["kernel"] and "head.tower0" in r["kernel"] and r["layer_frac"] == r["frac"]
    assert r["traffic"] is None and r["traffic_source"] is None          # another model than the PMC passes' -> no traffic figure
    # a direct-convolution launch: executed = algorithmic
    r = bench.tower_roofline(plan("head.tower0", 7, 1.0), 4 * ms, wl)
    assert r["flops_per_launch"] == alg and "direct" in r["flops_basis"] and "layer_frac" not in r
    # the split grid: frac describes the main launch (pro-rata FLOPs), layer_frac both launches with the full FLOPs
    share = 2040 / 2152
    p = plan("head.tower3x3", 16, 4.0, share=share, tail=True)
    med = [0.1, 0.83, 0.11, 0.2]
    layer = bench.tower_layer_ms(p, med)
    assert abs(layer - 0.94) < 1e-9
    r = bench.tower_roofline(p, 0.83, dict(bench.PMC_WORKLOAD), layer_ms=layer)
    assert abs(r["frac"] - alg * share / 4 / 0.83e-3 / 1e12 / bench.PEAK_F32_MFMA_TFLOPS) < 1e-3
    assert abs(r["layer_frac"] - alg / 4 / 0.94e-3 / 1e12 / bench.PEAK_F32_MFMA_TFLOPS) < 1e-3 and r["layer_frac"] < r["frac"]
    assert "two launches" in r["launch"]
    r = bench.tower_roofline(plan("head.tower3x3", 16, 4.0, sk=256), ms, dict(bench.PMC_WORKLOAD), layer_ms=ms)
    assert "persistent" in r["launch"] and "SK" in r["kernel"]
    # traffic: only for the PMC passes' own workload, and only for the kernel family they saw
    t, src = bench.pmc_traffic(dict(bench.PMC_WORKLOAD))
    rec = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    assert (t is not None) == os.path.exists(rec)
    for other in ({"model": "FCOS-B3"}, {"batch": 1}, {"height": 512, "width": 512}):
        assert bench.pmc_traffic({**bench.PMC_WORKLOAD, **other}) == (None, None)
    assert bench.pmc_traffic(dict(bench.PMC_WORKLOAD), "igemm") == (None, None)


def test_round5_entry_points_validate_on_the_host():
    """The round-5 additions are host arithmetic / host checks until a launch: workspace and pool sizes, weight packings, rejected argument combinations (no GPU needed)."""
    lib = _lib.lib()
    # persistent F(4x4) form: fixed 8 KB header + one 128 KB slot per workgroup; multiples of 8 up to 1024
    assert lib.fd_conv_sk_workspace_bytes(256) == 8192 + 256 * 16 * 512 * 16 and lib.fd_conv_sk_workspace_bytes(8) == 8192 + 8 * 131072
    assert lib.fd_conv_sk_workspace_bytes(12) == -1 and lib.fd_conv_sk_workspace_bytes(0) == -1 and lib.fd_conv_sk_workspace_bytes(2048) == -1
    p = _lib.ConvParams()
    p.x, p.w, p.y = 16, 16, 16
    p.segs = _lib.Segs.make(2, [(16, 16)])
    p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad, p.dil = 64, 64, 3, 3, 1, 1, 1
    p.x_cs, p.y_cs = 64, 64
    p.tile, p.sk_wgs = 4, 64
    assert lib.fd_conv2d_nhwc_f32(ctypes.byref(p), None) < 0 and b"WINOGRAD4" in lib.fd_last_error()         # sk_wgs on another tile
    p.tile, p.sk_wgs, p.io_f16 = _lib.WINO4_TILE, 0, 1
    assert lib.fd_conv2d_nhwc_f32(ctypes.byref(p), None) < 0 and b"io_f16" in lib.fd_last_error()            # f16 maps on a Winograd tile
    p.tile, p.io_f16 = 0, 2
    assert lib.fd_conv2d_nhwc_f32(ctypes.byref(p), None) < 0 and b"FD_PREC_F16" in lib.fd_last_error()       # f16 maps need the f16 arithmetic
    p.tile, p.io_f16, p.precision = _lib.F16K64_TILE, 0, 0
    assert lib.fd_conv2d_nhwc_f32(ctypes.byref(p), None) < 0 and b"F16K64" in lib.fd_last_error()            # the f16 kernel needs FD_PREC_F16
    # f16 K-tile-64 weight packing: reduction width % 64, never together with the (hi, lo) pair format
    assert lib.fd_pack_conv_weight_f32(ctypes.c_void_p(16), None, ctypes.c_void_p(16), 64, 96, 3, 3, 16, None) < 0
    assert lib.fd_pack_conv_weight_f32(ctypes.c_void_p(16), None, ctypes.c_void_p(16), 64, 64, 3, 3, 16 | 4, None) < 0
    w = _lib.WgradParams()
    w.io_f16 = 1
    assert lib.fd_conv2d_bwd_weight_f32(ctypes.byref(w), None) < 0
    # fused MBConv: tiles of 14 / 12 / 7 / 6 outputs per side
    assert lib.fd_mbconv_pool_bytes(16, 208, 336, 192, 3, 1) == 16 * 15 * 24 * 192 * 4 and lib.fd_mbconv_pool_bytes(2, 13, 25, 288, 5, 1) == 2 * 2 * 3 * 288 * 4
    assert lib.fd_mbconv_pool_bytes(2, 26, 42, 144, 3, 2) == 2 * 4 * 6 * 144 * 4 and lib.fd_mbconv_pool_bytes(1, 8, 8, 64, 7, 1) == -1
    assert lib.fd_mbconv_expand_dw_nhwc(None, 0, 0, None, None, None, None, None, None, None, 0, 0, None, 1, 8, 8, 24, 144, 3, 1, 1, 1, 8, 8, None) < 0
    assert ops.mbconv_fused_ok(24, 144, 3, 2) and ops.mbconv_fused_ok(48, 288, 5, 1) and not ops.mbconv_fused_ok(96, 576, 3, 1) and not ops.mbconv_fused_ok(20, 120, 3, 1)
    we = torch.arange(40 * 16, dtype=torch.float32).reshape(40, 16, 1, 1)        # mid = 40 (padded to 64), Cin = 16: element (cb, g, h, l, jj) = w[32 cb + l][8 h + 4 g + jj]
    pk = ops.pack_mbconv_expand_weight(we)
    assert pk.shape == (2, 2, 2, 32, 4)
    assert float(pk[0, 1, 1, 5, 2]) == float(we[5, 8 + 4 + 2, 0, 0]) and float(pk[1, 0, 0, 7, 3]) == float(we[39, 3, 0, 0]) and float(pk[1, 1, 1, 8, 0]) == 0.0
    # AMP packing choice
    from pytorch_object_detection_amd import train_ops as T
    assert T.amp_pack(0, 256, 256) is False and T.amp_pack(_lib.PREC_F16, 256, 64) == 2 and T.amp_pack(_lib.PREC_F16, 32, 256) is True
    assert ops.f16k64_ok(128, 80) and not ops.f16k64_ok(96, 64)


def test_graphed_step_refuses_host_tensors():
    """train_graph.GraphedStep records a training step as a HIP graph: without CUDA tensors there is nothing to record -- it must say so before touching a stream."""
    import torch
    from pytorch_object_detection_amd.train_graph import GraphedStep
    with pytest.raises(ValueError, match="CUDA tensors"):
        GraphedStep(lambda x: x.sum(), [torch.zeros(2, 3)])
    with pytest.raises(ValueError, match="CUDA tensors"):
        GraphedStep(lambda: torch.zeros(()), [])
