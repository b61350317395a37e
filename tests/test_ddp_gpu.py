"""Cfg4 multi-rank path: two processes (both on cuda:0 -- the box has one GPU -- over the gloo backend, which carries CUDA
tensors through the host) train one step under DistributedDataParallel with different images per rank; the all-reduced
gradients must equal the gradients of ONE process on the concatenated batch (FCOSLoss is a mean over images, frozen BN
and GroupNorm are per-sample, so the average of the ranks' gradients is the gradient of the full batch)."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(dev):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 128, 160, generator=g)
    c = torch.rand(4, 3, 2, generator=g) * 60 + 30
    s = torch.rand(4, 3, 2, generator=g) * 50 + 12
    gt = torch.cat([c - s / 2, c + s / 2], -1)
    labels = torch.randint(1, 21, (4, 3), generator=g)
    return x.to(dev), gt.to(dev), labels.to(dev)


def _model(dev):
    from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
    torch.manual_seed(3)
    m = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).to(dev)
    m.freeze_all_bn = True     # batch-statistic FPN BatchNorms would need SyncBatchNorm (train.py:103) for rank-sum == full batch
    return m.train()


def _step(net, x, gt, labels):
    from pytorch_object_detection_amd.model.loss import FCOSLoss
    from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
    gen = FCOSGenTargets([8, 16, 32, 64, 128], [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]])
    out = net(x)
    loss = FCOSLoss("giou")([out, gen([out, gt, labels])])[-1]
    loss.mean().backward()
    return float(loss.detach())


NAMES = ("head.cls_logits.weight", "head.reg_conv.0.weight", "fpn.tf1.weight", "fpn.HisBlock3.conv4.weight",
         "backbone.extract_feature.layer4.2.conv3.weight", "backbone.extract_feature.layer2.0.conv1.weight")


def _worker(rank, world, init_file, out_dir):
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        model = _model(dev)
        net = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)      # as train.py:101
        x, gt, labels = _data(dev)
        lo, hi = rank * 2, rank * 2 + 2
        loss = _step(net, x[lo:hi], gt[lo:hi], labels[lo:hi])
        params = dict(model.named_parameters())
        torch.save({"loss": loss, **{n: params[n].grad.cpu() for n in NAMES}}, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_gradients_equal_the_full_batch():
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rdzv")
        mp.spawn(_worker, args=(2, init_file, tmp), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(tmp, "rank0.pt")), torch.load(os.path.join(tmp, "rank1.pt"))
    dev = torch.device("cuda", 0)
    model = _model(dev)
    x, gt, labels = _data(dev)
    full_loss = _step(model, x, gt, labels)
    params = dict(model.named_parameters())
    assert abs((r0["loss"] + r1["loss"]) / 2 - full_loss) < 2e-4 * abs(full_loss)
    for n in NAMES:
        assert torch.equal(r0[n], r1[n]), n                        # every rank holds the all-reduced gradient
        ref = params[n].grad.cpu()
        scale = float(ref.abs().max()) + 1e-12
        d = ((r0[n] - ref).abs() / scale).flatten()
        # bulk at rounding level; a ReLU-mask element flipping between the 2-image and the 4-image plans may move a few entries
        assert float(d.median()) < 1e-4 and float((d > 2e-2).float().mean()) < 0.01, (n, float(d.median()), float(d.max()))
