"""World-size-2 checks of the multi-GPU host logic on CPU (gloo): inference shards by contiguous
slices with no data-path collective; only final keypoints are all-gathered; the training exchange
step is a bucketed gradient all-reduce (mean)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from udp_pose_amd.dist import allreduce_mean_, allreduce_sum_, gather_keypoints, shard_bounds


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 65, 1000):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_total * 17 * 3, dtype=torch.float32).reshape(n_total, 17, 3)
        lo, hi = shard_bounds(n_total, rank, world)
        got = gather_keypoints(full[lo:hi].clone(), n_total)
        ok_gather = bool(torch.equal(got, full))
        g = torch.full((1000,), float(rank + 1))
        allreduce_mean_(g, bucket_elems=300)
        ok_reduce = bool(torch.allclose(g, torch.full((1000,), (1 + world) / 2.0)))
        g = torch.full((1000,), float(rank + 1))
        allreduce_sum_(g, bucket_elems=256)
        ok_reduce &= bool(torch.equal(g, torch.full((1000,), float(sum(range(1, world + 1))))))
        q.put((rank, ok_gather, ok_reduce))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 64])
def test_two_rank_gather_and_allreduce(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res)
