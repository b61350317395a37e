"""world_size-2 gloo test of the detection all-gather (the only collective of the sharded inference path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytorch_object_detection_amd.dist import gather_detections, pack_detections, shard_batch, unpack_detections, unpad_gathered


def test_shard_batch_partition():
    for g, w in ((128, 8), (16, 1), (10, 4), (3, 8)):
        spans = [shard_batch(g, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == g
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_pack_roundtrip():
    s, c, b = torch.rand(3, 7), torch.randint(1, 81, (3, 7)), torch.rand(3, 7, 4) * 640
    n = torch.tensor([7, 0, 3], dtype=torch.int32)
    rec = pack_detections(s, c, b, n)
    assert rec.shape == (3, 8, 6) and rec.dtype == torch.float32      # ONE message: counts ride in row 0
    s2, c2, b2, n2 = unpack_detections(rec)
    assert torch.equal(s, s2) and torch.equal(c, c2) and torch.equal(b, b2) and torch.equal(n, n2) and n2.dtype == torch.int32


def test_gather_is_a_single_collective(monkeypatch):
    """north_star: 'a single RCCL all-gather of final detections' -- count the collectives gather_detections issues."""
    calls = []
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "all_gather_into_tensor", lambda out, inp, group=None: (calls.append(tuple(inp.shape)), out.copy_(torch.cat([inp, inp])))[0])
    s, c, b, n = torch.rand(2, 5), torch.randint(1, 81, (2, 5)), torch.rand(2, 5, 4), torch.tensor([5, 2], dtype=torch.int32)
    gs, gc, gb, gn = gather_detections(s, c, b, n)
    assert calls == [(2, 6, 6)] and gn.tolist() == [5, 2, 5, 2] and torch.equal(gs[2:], s)


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gen = torch.Generator().manual_seed(100 + rank)
        B, K = 2, 5
        s = torch.rand(B, K, generator=gen)
        c = torch.randint(1, 81, (B, K), generator=gen)
        b = torch.rand(B, K, 4, generator=gen) * 640
        n = torch.tensor([rank + 1, K - rank], dtype=torch.int32)
        gs, gc, gb, gn = gather_detections(s, c, b, n)
        assert gs.shape == (world * B, K) and gb.shape == (world * B, K, 4) and gn.shape == (world * B,)
        for r in range(world):
            gen_r = torch.Generator().manual_seed(100 + r)
            es = torch.rand(B, K, generator=gen_r)
            ec = torch.randint(1, 81, (B, K), generator=gen_r)
            eb = torch.rand(B, K, 4, generator=gen_r) * 640
            assert torch.equal(gs[r * B:(r + 1) * B], es) and torch.equal(gc[r * B:(r + 1) * B], ec)
            assert torch.equal(gb[r * B:(r + 1) * B], eb)
            assert gn[r * B:(r + 1) * B].tolist() == [r + 1, K - r]
    finally:
        dist.destroy_process_group()


def test_gather_detections_gloo_world2():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)


def _worker_uneven(rank, world, port):
    """Global batch 5 over 2 ranks: shards of 3 and 2 images -- the short shard is padded to the largest (pad_to), the padding dropped after."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G, K = 5, 4
        gen = torch.Generator().manual_seed(7)
        S, Cc, Bx = torch.rand(G, K, generator=gen), torch.randint(1, 81, (G, K), generator=gen), torch.rand(G, K, 4, generator=gen) * 640
        N = torch.tensor([4, 0, 2, 1, 3], dtype=torch.int32)
        lo, hi = shard_batch(G, rank, world)
        pad = max(h - l for l, h in (shard_batch(G, r, world) for r in range(world)))
        got = unpad_gathered(gather_detections(S[lo:hi], Cc[lo:hi], Bx[lo:hi], N[lo:hi], pad_to=pad), G, world)
        assert torch.equal(got[0], S) and torch.equal(got[1], Cc) and torch.equal(got[2], Bx) and torch.equal(got[3], N)
    finally:
        dist.destroy_process_group()


def test_gather_detections_uneven_shards_gloo_world2():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker_uneven, args=(2, port), nprocs=2, join=True)


# ---------------------------------------------------------------------------------------------- bench.py's own launcher
def _bench(args, env_extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


@pytest.mark.parametrize("mode", ["infer", "train"])
def test_bench_gpus_n_launches_n_ranks_itself(mode):
    """`python bench.py --gpus N` with no launcher around it starts N rank processes, relays rank 0's ONE line and reports
    n_gpus == N (it used to run one rank silently).  FD_BENCH_DRYRUN: the ranks rendezvous over gloo and run a
    collective but no GPU work, so the control flow is checked on a CPU-only host."""
    rc, lines, err = _bench(["--gpus", "2", "--mode", mode], {"FD_BENCH_DRYRUN": "1"})
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["dryrun"] is True and lines[0]["mode"] == mode


def test_bench_launcher_fails_when_a_rank_fails():
    """No GPU here: every rank exits non-zero ('bench.py needs a GPU'); the launcher must say so and print no line."""
    if torch.cuda.is_available():
        pytest.skip("needs a CPU-only host")
    rc, lines, err = _bench(["--gpus", "2"], {})
    assert rc != 0 and lines == [] and "exited with code" in err


def test_bench_rejects_world_size_mismatch():
    rc, lines, err = _bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "FD_BENCH_DRYRUN": "1"})
    assert rc != 0 and lines == [] and "--gpus 2 but WORLD_SIZE=1" in err
