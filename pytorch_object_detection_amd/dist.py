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


def pack_detections(scores: torch.Tensor, classes: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
    """[B,K] f32, [B,K] i64, [B,K,4] f32 -> [B,K,6] f32 rows (x1, y1, x2, y2, score, class); class ids are < 2^24."""
    return torch.cat([boxes, scores.unsqueeze(-1), classes.to(torch.float32).unsqueeze(-1)], dim=-1).contiguous()


def unpack_detections(packed: torch.Tensor):
    return packed[..., 4].contiguous(), packed[..., 5].to(torch.int64), packed[..., :4].contiguous()


def gather_detections(scores, classes, boxes, counts, group=None, force: bool = False):
    """All-gather padded detections from every rank: returns (scores [W*B,K], classes, boxes [W*B,K,4], counts [W*B]).
    Every rank must hold the same local batch B and K (pad the last shard).  Two fixed-size messages per rank:
    B*K*6 floats (384 KB at B=16, K=1000) and B int32 counts."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return scores, classes, boxes, counts
    world = dist.get_world_size(group)
    packed = pack_detections(scores, classes, boxes)
    out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    cnt = torch.empty(world * counts.shape[0], dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    dist.all_gather_into_tensor(cnt, counts.contiguous(), group=group)
    s, c, b = unpack_detections(out)
    return s, c, b, cnt
