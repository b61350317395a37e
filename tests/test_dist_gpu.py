"""Cfg3 multi-rank path on the one-GPU box: the detection all-gather over RCCL with world size 1 (force=True exercises the
real ncclAllGather and the HIP pack / unpack launches), and with TWO processes (both on cuda:0, the gloo backend carrying
the CUDA records through the host — RCCL refuses two ranks on one device) running the sharded inference step on different
image shards; the gathered result must equal one process's detections on the whole batch."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_pack_unpack_kernels_match_the_torch_statement():
    from pytorch_object_detection_amd import dist as D
    g = torch.Generator().manual_seed(0)
    s, c, b = torch.rand(5, 1000, generator=g), torch.randint(1, 81, (5, 1000), generator=g), torch.rand(5, 1000, 4, generator=g) * 640
    n = torch.tensor([1000, 0, 17, 999, 1], dtype=torch.int32)
    rec_cpu = D.pack_detections(s, c, b, n)
    rec = D.pack_detections(s.to(DEV), c.to(DEV), b.to(DEV), n.to(DEV))
    assert rec.shape == (5, 1001, 6) and torch.equal(rec.cpu(), rec_cpu)
    s2, c2, b2, n2 = D.unpack_detections(rec)
    assert torch.equal(s2.cpu(), s) and torch.equal(c2.cpu(), c) and torch.equal(b2.cpu(), b) and torch.equal(n2.cpu(), n)


def test_gather_detections_over_rccl_world1():
    import torch.distributed as dist
    from pytorch_object_detection_amd.dist import gather_detections
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        g = torch.Generator().manual_seed(1)
        s, c, b = torch.rand(16, 1000, generator=g).to(DEV), torch.randint(1, 81, (16, 1000), generator=g).to(DEV), \
            (torch.rand(16, 1000, 4, generator=g) * 640).to(DEV)
        n = torch.randint(0, 1001, (16,), generator=g).to(torch.int32).to(DEV)
        calls = []
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            gs, gc, gb, gn = gather_detections(s, c, b, n, force=True)
        finally:
            dist.all_gather_into_tensor = real
        assert len(calls) == 1                                      # ONE collective (north_star)
        assert torch.equal(gs, s) and torch.equal(gc, c) and torch.equal(gb, b) and torch.equal(gn, n)
        assert gather_detections(s, c, b, n)[0] is s                # world size 1 without force: no collective at all
    finally:
        dist.destroy_process_group()


def _model(dev):
    from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
    from test_model_gpu import randomize_norms
    torch.manual_seed(3)
    m = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).eval()
    randomize_norms(m, 4)
    return m.to(dev)


def _detect(model, x):
    from pytorch_object_detection_amd.model.modules.head import ClipBoxes, FCOSHead
    head = FCOSHead(0.05, 0.6, 1000, [8, 16, 32, 64, 128])
    s, c, b, n = head.detect_padded(model(x))
    return s, c, ClipBoxes()(x, b), n


def _worker(rank, world, init_file, out_dir):
    import torch.distributed as dist
    from pytorch_object_detection_amd.dist import gather_detections, shard_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        x = torch.randn(4, 3, 128, 160, generator=torch.Generator().manual_seed(7)).to(dev)
        lo, hi = shard_batch(4, rank, world)
        res = gather_detections(*_detect(_model(dev), x[lo:hi].contiguous()))
        torch.save([t.cpu() for t in res], os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_inference_equals_the_whole_batch():
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, os.path.join(tmp, "rdzv"), tmp), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(tmp, "rank0.pt")), torch.load(os.path.join(tmp, "rank1.pt"))
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)                                    # every rank holds all detections
    x = torch.randn(4, 3, 128, 160, generator=torch.Generator().manual_seed(7)).to(DEV)
    model = _model(torch.device(DEV))
    parts = [_detect(model, x[i:i + 2].contiguous()) for i in (0, 2)]
    for k in range(4):
        np.testing.assert_array_equal(r0[k].numpy(), torch.cat([p[k] for p in parts]).cpu().numpy())
