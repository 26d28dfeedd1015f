"""The N > 1 path on CPU: world_size-2 gloo run of the keypoint all-gather (the only collective of the
data-parallel path; RCCL on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from balf_amd import pipeline


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, b, k, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = pipeline.shard_range(world * b, rank, world)
        g = torch.Generator().manual_seed(1234)
        all_idx = torch.randint(0, 1 << 20, (world * b, k), generator=g, dtype=torch.int32)
        all_sc = torch.rand((world * b, k), generator=g)
        all_cnt = torch.randint(0, k + 1, (world * b,), generator=g, dtype=torch.int32)
        idx, sc, cnt = pipeline.allgather_keypoints(all_idx[lo:hi].clone(), all_sc[lo:hi].clone(), all_cnt[lo:hi].clone())
        ok = torch.equal(idx, all_idx) and torch.equal(sc, all_sc) and torch.equal(cnt, all_cnt)
        out_q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_allgather_keypoints_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 3, 17, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def _uneven_worker(rank, world, port, total, k, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = pipeline.shard_range(total, rank, world)
        g = torch.Generator().manual_seed(99)
        all_idx = torch.randint(0, 1 << 20, (total, k), generator=g, dtype=torch.int32)
        all_sc = torch.rand((total, k), generator=g)
        all_cnt = torch.randint(0, k + 1, (total,), generator=g, dtype=torch.int32)
        idx, sc, cnt = pipeline.allgather_keypoints(all_idx[lo:hi].clone(), all_sc[lo:hi].clone(), all_cnt[lo:hi].clone(), total=total)
        ok = torch.equal(idx, all_idx) and torch.equal(sc, all_sc) and torch.equal(cnt, all_cnt)
        # without total= the unequal shards must be refused on EVERY rank (not hang, not gather garbage)
        refused = False
        try:
            pipeline.allgather_keypoints(all_idx[lo:hi].clone(), all_sc[lo:hi].clone(), all_cnt[lo:hi].clone())
        except ValueError:
            refused = True
        out_q.put((rank, bool(ok), refused))
    finally:
        dist.destroy_process_group()


def _run_uneven(world, total):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, total, 11, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    uneven = total % world != 0
    assert res == [(r, True, uneven) for r in range(world)], res


def test_allgather_unequal_shards_world2_b5():
    """A batch that does not divide by the world size: shards of 3 + 2 images, padded to 3 rows for the
    collective, the padding stripped again."""
    _run_uneven(2, 5)


def test_allgather_unequal_shards_world3_b7():
    _run_uneven(3, 7)


def test_allgather_unequal_shards_world3_b2():
    """Fewer images than ranks: one rank holds nothing."""
    _run_uneven(3, 2)


def test_allgather_is_identity_without_process_group():
    idx = torch.zeros((2, 5), dtype=torch.int32); sc = torch.zeros((2, 5)); cnt = torch.zeros(2, dtype=torch.int32)
    a, b, c = pipeline.allgather_keypoints(idx, sc, cnt)
    assert a is idx and b is sc and c is cnt


def _single_worker(port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        idx = torch.arange(10, dtype=torch.int32).view(2, 5); sc = torch.rand((2, 5)); cnt = torch.tensor([5, 3], dtype=torch.int32)
        a, b, c = pipeline.allgather_keypoints(idx, sc, cnt)                 # single rank: shortcut
        short = a is idx
        a, b, c = pipeline.allgather_keypoints(idx, sc, cnt, force=True)     # forced: the collective runs
        out_q.put(bool(short and a is not idx and torch.equal(a, idx) and torch.equal(b, sc) and torch.equal(c, cnt)))
    finally:
        dist.destroy_process_group()


def test_allgather_single_rank_forced():
    """world size 1 with ``force=True`` goes through pack -> all_gather_into_tensor -> unpack (the RCCL plumbing
    test of the GPU box, tests/test_rccl_gpu.py, on gloo)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
    assert p.exitcode == 0
