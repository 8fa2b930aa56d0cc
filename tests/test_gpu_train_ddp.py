"""Two-rank training step on the GPU box: one process per rank (both on the single
GPU of the test box, gloo transport -- RCCL needs distinct devices; bench/production use backend nccl),
each rank runs its shard through HRNetTrainer.train_step(world_size=2).

Checks: both ranks end with bit-identical parameters; those equal a single-process emulation that
computes the two shard gradients one after the other, sums them and applies Adam with grad_scale 1/2
(BatchNorm statistics stay per rank, as in the reference's nn.DataParallel replicas)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from udp_pose_amd import synth                      # noqa: E402

EXTRA = synth.scaled_extra(32, modules=(1, 1, 1), blocks=1)
CFG = {"MODEL": {"EXTRA": EXTRA, "NUM_JOINTS": 17, "TARGET_TYPE": "gaussian"}}


def _batch(rank):
    x = torch.from_numpy(synth.synth_crops(4, 128, 96, seed=30 + rank))
    tg = torch.from_numpy(synth.synth_heatmaps(4, 17, 32, 24, seed=40 + rank))
    return x, tg, torch.ones(4, 17, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, graphed=False, steps=2, vary=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from udp_pose_amd.train import HRNetTrainer
        tr = HRNetTrainer(CFG, synth.synth_state_dict(EXTRA, 17, "gaussian", seed=2), device="cuda")
        x, tg, tw = _batch(rank)
        nseg = 0
        for k in range(steps):
            xk = (x + 0.01 * k if vary else x).cuda()      # vary: a different batch every step
            if graphed:
                loss = tr.train_step_graphed(xk, tg.cuda(), tw.cuda(), world_size=world)
            else:
                loss = tr.train_step(xk, tg.cuda(), tw.cuda(), world_size=world)
        torch.cuda.synchronize()
        if graphed:
            nseg = max(len(v[0]) for v in tr._graphs.values())
        q.put((rank, tr.flat[:tr._n_param].cpu().numpy(), float(loss.cpu()[0]), nseg, list(tr.reduce_order), len(tr._buckets)))
    finally:
        dist.destroy_process_group()


def _worker_w32(rank, world, port, q, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    try:
        from udp_pose_amd.train import HRNetTrainer
        cfg = {"MODEL": {"EXTRA": synth.W32_EXTRA, "NUM_JOINTS": 17, "TARGET_TYPE": "gaussian"}}
        tr = HRNetTrainer(cfg, synth.synth_state_dict(synth.W32_EXTRA, 17, "gaussian", seed=0), device=dev)
        n = 32
        x = torch.from_numpy(np.tile(synth.synth_crops(8, 256, 192, seed=50 + rank), (4, 1, 1, 1)))
        tg = torch.from_numpy(synth.synth_heatmaps(n, 17, 64, 48, seed=60 + rank))
        for _ in range(2):
            loss = tr.train_step(x.to(dev), tg.to(dev), torch.ones(n, 17, 1, device=dev), world_size=world)
        torch.cuda.synchronize()
        p = tr.flat[:tr._n_param]
        q.put((rank, bool(torch.isfinite(p).all()), float(p.double().sum()), float((p.double() ** 2).sum()),
               p[::997].cpu().numpy(), float(loss.cpu()[0]), list(tr.reduce_order), len(tr._buckets)))
    finally:
        dist.destroy_process_group()


def test_two_rank_w32_config3_size_replicas_stay_identical():
    """Config 3 at its per-GPU size (W32 256x192, 32 images per rank), two ranks, two steps through the SUM
    all-reduce of the flat 28.6 M-element gradient: parameters finite and identical on both ranks.  With two
    visible GPUs the transport is RCCL (backend nccl, one device per rank); on the one-GPU test box both ranks
    share the device and gloo carries the all-reduce."""
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_w32, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]
    np.testing.assert_array_equal(res[0][4], res[1][4])
    assert np.isfinite(res[0][5]) and np.isfinite(res[1][5])
    # the exchange is overlapped with the backward: every ~25 MB bucket is all-reduced exactly once, issued as soon
    # as its last gradient is written -- the bucket of the LAST layers first, the stem's bucket last
    order, nb = res[0][6], res[0][7]
    assert nb >= 4 and sorted(order) == list(range(nb)) and order == res[1][6]
    assert order[0] == nb - 1 and order[-1] == 0, order


def test_two_rank_train_step_equals_summed_shard_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])          # replicas stay in lock-step
    # single-process emulation
    from udp_pose_amd import _lib
    from udp_pose_amd.train import HRNetTrainer
    sd = synth.synth_state_dict(EXTRA, 17, "gaussian", seed=2)
    master = HRNetTrainer(CFG, sd, device="cuda")
    shards = [HRNetTrainer(CFG, sd, device="cuda") for _ in range(2)]
    for _ in range(2):
        total = torch.zeros_like(master.grad)
        for r, tr in enumerate(shards):
            tr.flat[:tr._n_param].copy_(master.flat[:master._n_param])      # parameters are replicated
            x, tg, tw = _batch(r)
            heat = tr.forward(x.cuda())
            _, d = tr.loss_and_grad(heat, tg.cuda(), tw.cuda())
            tr.backward(d)
            total += tr.grad
        master.grad.copy_(total)
        master.adam_step(0.5)
    np.testing.assert_allclose(res[0][1], master.flat[:master._n_param].cpu().numpy(), rtol=0, atol=1e-6)


def _run_two(graphed, steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, graphed, steps, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_rank_graphed_step_equals_eager_two_rank_step():
    """the multi-rank step replayed as hipGraph segments (cut where a gradient bucket is complete;
    the host issues the bucket's all-reduce and replays the next segment at once) leaves BOTH ranks with the parameters
    of the eager two-rank path, bit for bit, after five steps on changing batches (step 1 eager, step 2 captures, steps
    3-5 replay), reduces every bucket exactly once per step and in the same order."""
    eager = _run_two(False, 5)
    graph = _run_two(True, 5)
    np.testing.assert_array_equal(graph[0][1], graph[1][1])
    np.testing.assert_array_equal(graph[0][1], eager[0][1])
    assert abs(graph[0][2] - eager[0][2]) <= 1e-12 * abs(eager[0][2])     # (the loss is an fp64 atomic sum: last bits vary)
    nb = graph[0][5]
    assert graph[0][3] >= 3                              # at least one backward segment + the wait marker + Adam
    assert sorted(graph[0][4]) == list(range(nb)) and graph[0][4] == eager[0][4] == graph[1][4]
