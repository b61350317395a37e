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


def _model(dev, freeze_all_bn=True):
    from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
    torch.manual_seed(3)
    m = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).to(dev)
    m.freeze_all_bn = freeze_all_bn     # False = the reference's mode: FPN BatchNorms on batch statistics (SyncBatchNorm under DDP, train.py:103)
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


def _worker(rank, world, init_file, out_dir, sync_bn=False):
    import torch.distributed as dist
    from pytorch_object_detection_amd import train_ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        model = _model(dev, freeze_all_bn=not sync_bn)
        net = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)      # as train.py:101
        if sync_bn:
            net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)                               # as train.py:103
            n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) and m.training for m in net.modules())
            assert n_sync >= 30, n_sync                                                            # the FPN's batch-statistic BatchNorms
            calls = []
            real = dist.all_reduce
            dist.all_reduce = lambda t, *a, **k: (calls.append((t.dtype, t.numel())), real(t, *a, **k))[1]
        x, gt, labels = _data(dev)
        lo, hi = rank * 2, rank * 2 + 2
        loss = _step(net, x[lo:hi], gt[lo:hi], labels[lo:hi])
        params = dict(model.named_parameters())
        out = {"loss": loss, **{n: params[n].grad.cpu() for n in NAMES}}
        if sync_bn:
            dist.all_reduce = real
            assert train_ops.STATS["stock_fallbacks"] == 0                   # no layer of the step left the HIP kernels (FD_STRICT would have raised)
            stat_calls = [c for c in calls if c[0] == torch.float64]
            assert len(stat_calls) >= 2 * 30, len(stat_calls)                # one fp64 all-reduce per SyncBatchNorm forward and one per backward
            # the collectives are issued asynchronously and waited for where their result is first needed: SYNC_TRACE holds, per
            # collective in issue order, the number of HIP launches enqueued between its issue and its wait -- independent work under the all-reduce
            # (HisBlock: conv2 beside bn1, the SE branch beside bn2, forward and -- mirrored by autograd's node order -- backward; the FPN laterals)
            tr = list(train_ops.SYNC_TRACE)
            assert len(tr) == len(stat_calls), (len(tr), len(stat_calls))
            fwd, bwd = [d for k, d in tr if k == "fwd"], [d for k, d in tr if k == "bwd"]
            assert len(fwd) == len(bwd) >= 30
            over = sum(d > 0 for d in fwd) + sum(d > 0 for d in bwd)
            out["sync_overlapped"], out["sync_total"] = over, len(tr)
            assert 2 * over >= len(tr), (sum(d > 0 for d in fwd), sum(d > 0 for d in bwd), len(tr))
            assert not train_ops._PENDING                                  # every deferred input gradient was finished by its consumer
            out.update({k: v.cpu() for k, v in model.state_dict().items() if "fpn" in k and ("running" in k or "num_batches" in k)})
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
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


def test_two_rank_syncbatchnorm_on_hip_equals_the_full_batch():
    """The reference's training arithmetic under DDP (train.py:101-103,151): model wrapped in DistributedDataParallel, converted with
    SyncBatchNorm.convert_sync_batchnorm, FPN BatchNorms on batch statistics.  The converted layers run on the HIP statistics kernels with
    one fp64 all-reduce per direction (train_ops._SyncBatchNormTrainRows), so two ranks x 2 images must reproduce ONE process on the 4
    images with plain BatchNorm2d: loss, gradients, running statistics, num_batches_tracked."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rdzv")
        mp.spawn(_worker, args=(2, init_file, tmp, True), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(tmp, "rank0.pt")), torch.load(os.path.join(tmp, "rank1.pt"))
    dev = torch.device("cuda", 0)
    model = _model(dev, freeze_all_bn=False)
    x, gt, labels = _data(dev)
    full_loss = _step(model, x, gt, labels)
    params = dict(model.named_parameters())
    assert abs((r0["loss"] + r1["loss"]) / 2 - full_loss) < 5e-4 * abs(full_loss), (r0["loss"], r1["loss"], full_loss)
    sd = model.state_dict()
    nstat = 0
    for k, v in r0.items():
        if "running" in k or "num_batches" in k:
            assert torch.equal(v, r1[k]), k                            # both ranks normalised with the same global statistics
            np.testing.assert_allclose(v.numpy(), sd[k].cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
            nstat += 1
    assert nstat >= 90 and int(sd["fpn.gn1.num_batches_tracked"]) == 1
    for n in NAMES:
        assert torch.equal(r0[n], r1[n]), n
        ref = params[n].grad.cpu()
        scale = float(ref.abs().max()) + 1e-12
        d = ((r0[n] - ref).abs() / scale).flatten()
        # gradients through ~30 batch-statistic BatchNorms are ill-conditioned (tests/test_train_gpu.py): bulk at 1e-3 of the maximum
        assert float(d.median()) < 2e-3 and float((d > 5e-2).float().mean()) < 0.02, (n, float(d.median()), float(d.max()))


def _uneven_rows(rank_or_all):
    """Rows of a 32-channel map per rank: rank 0 holds 2 x 16 x 16, rank 1 holds 2 x 16 x 24 -- what dataset/voc.py:141-171 produces when the
    ranks' batches are padded to different H x W."""
    g = torch.Generator().manual_seed(77)
    parts = [torch.randn(2 * 16 * 16, 32, generator=g) * 1.5 + 0.3, torch.randn(2 * 16 * 24, 32, generator=g) * 0.7 - 0.2]
    grads = [torch.randn(p.shape, generator=g) for p in parts]
    return (parts, grads) if rank_or_all is None else (parts[rank_or_all], grads[rank_or_all])


def _uneven_worker(rank, world, init_file, out_dir):
    import torch.distributed as dist
    from pytorch_object_detection_amd import train_ops
    from pytorch_object_detection_amd._lib import ACT_SILU
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        torch.manual_seed(9)
        bn = torch.nn.SyncBatchNorm(32).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(32) + 0.5); bn.bias.copy_(torch.randn(32) * 0.1)
        x, gy = _uneven_rows(rank)
        x = x.to(dev).requires_grad_(True)
        y = train_ops.batchnorm_train_rows(bn, x, ACT_SILU)
        y.backward(gy.to(dev))
        torch.save({"y": y.detach().cpu(), "dx": x.grad.cpu(), "dgamma": bn.weight.grad.cpu(), "dbeta": bn.bias.grad.cpu(),
                    "rm": bn.running_mean.cpu(), "rv": bn.running_var.cpu(), "nbt": int(bn.num_batches_tracked)}, os.path.join(out_dir, f"u{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_syncbatchnorm_with_different_row_counts_per_rank():
    """ranks whose batches are padded to different H x W hold different row counts; the global count must come out of the
    all-reduce (read on the device), not rows * world_size.  Two ranks with 512 and 768 rows against plain BatchNorm over the 1280 rows."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_uneven_worker, args=(2, os.path.join(tmp, "rdzv"), tmp), nprocs=2, join=True)
        r = [torch.load(os.path.join(tmp, f"u{i}.pt")) for i in range(2)]
    parts, grads = _uneven_rows(None)
    torch.manual_seed(9)
    bn = torch.nn.BatchNorm1d(32).double().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(32) + 0.5); bn.bias.copy_(torch.randn(32) * 0.1)
    x = torch.cat(parts).double().requires_grad_(True)
    y = torch.nn.functional.silu(bn(x))
    y.backward(torch.cat(grads).double())
    n0 = parts[0].shape[0]
    for i, sl in enumerate((slice(0, n0), slice(n0, None))):
        np.testing.assert_allclose(r[i]["y"].numpy(), y[sl].detach().float().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(r[i]["dx"].numpy(), x.grad[sl].float().numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(r[i]["rm"].numpy(), bn.running_mean.float().numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(r[i]["rv"].numpy(), bn.running_var.float().numpy(), rtol=1e-5, atol=1e-7)
        assert r[i]["nbt"] == 1
    # the affine gradients are per-rank sums (DDP averages them): together they are the full batch's
    np.testing.assert_allclose((r[0]["dgamma"] + r[1]["dgamma"]).numpy(), bn.weight.grad.float().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose((r[0]["dbeta"] + r[1]["dbeta"]).numpy(), bn.bias.grad.float().numpy(), rtol=2e-4, atol=2e-5)
