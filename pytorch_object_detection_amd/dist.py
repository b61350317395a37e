"""Image-sharded multi-GPU inference: one process per GPU (torch.distributed, backend 'nccl' = RCCL over xGMI),
batch split in contiguous slices, weights replicated, and ONE collective — an all-gather of the fixed-size padded
detections.  The network itself exchanges nothing (frozen BN, per-sample GN / SE)."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_batch(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_detections(scores: torch.Tensor, classes: torch.Tensor, boxes: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """[B,K] f32, [B,K] i64, [B,K,4] f32, [B] i32 -> ONE message [B, K+1, 6] f32: row 0 of an image carries its count, rows
    1..K are (x1, y1, x2, y2, score, class); class ids and counts are < 2^24, exact in fp32.  One HIP launch on CUDA
    tensors (fd_pack_detections); CPU tensors (the gloo rehearsal of the protocol in tests/test_dist_cpu.py) take the
    equivalent torch ops."""
    if scores.is_cuda:
        from . import ops
        return ops.pack_detections(scores, classes, boxes, counts)
    B, K = scores.shape
    rec = torch.zeros(B, K + 1, 6, dtype=torch.float32)
    rec[:, 0, 0] = counts.to(torch.float32)
    rec[:, 1:, :4], rec[:, 1:, 4], rec[:, 1:, 5] = boxes, scores, classes.to(torch.float32)
    return rec


def unpack_detections(rec: torch.Tensor):
    """Inverse of pack_detections: -> (scores [B,K], classes [B,K] int64, boxes [B,K,4], counts [B] int32)."""
    if rec.is_cuda:
        from . import ops
        return ops.unpack_detections(rec)
    return (rec[:, 1:, 4].contiguous(), rec[:, 1:, 5].to(torch.int64), rec[:, 1:, :4].contiguous(), rec[:, 0, 0].to(torch.int32))


def gather_detections(scores, classes, boxes, counts, group=None, force: bool = False, pad_to: int = 0):
    """All-gather padded detections from every rank: returns (scores [W*B,K], classes, boxes [W*B,K,4], counts [W*B]).
    ONE collective: a single all_gather_into_tensor of the [B, K+1, 6] fp32 records (385 KB at B=16, K=1000; latency-bound on xGMI),
    one pack launch before it and one unpack launch after it.  Every rank must contribute the same number of records: with equal
    shards (the bench: 16 images per GPU) that is the local batch; for a global batch that does not divide (shard_batch hands the
    first ranks one image more) pass pad_to = the LARGEST local batch -- shorter shards are padded with empty records (count 0) and
    the caller drops them with `unpad_gathered` (the shard sizes are known analytically: no second collective)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return scores, classes, boxes, counts
    world = dist.get_world_size(group)
    rec = pack_detections(scores, classes, boxes, counts)
    if pad_to and pad_to > rec.shape[0]:
        rec = torch.cat([rec, torch.zeros((pad_to - rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)], 0)
    out = torch.empty((world * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec, group=group)
    return unpack_detections(out)


def unpad_gathered(gathered, global_batch: int, world: int):
    """Drop the padding records gather_detections(pad_to=...) added: rank r contributed shard_batch(global_batch, r, world) images."""
    sizes = [hi - lo for lo, hi in (shard_batch(global_batch, r, world) for r in range(world))]
    pad = max(sizes)
    idx = torch.cat([torch.arange(r * pad, r * pad + n) for r, n in enumerate(sizes)]).to(gathered[0].device)
    return tuple(t.index_select(0, idx) for t in gathered)
