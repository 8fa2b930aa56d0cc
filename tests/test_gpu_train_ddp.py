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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from udp_pose_amd.train import HRNetTrainer
        tr = HRNetTrainer(CFG, synth.synth_state_dict(EXTRA, 17, "gaussian", seed=2), device="cuda")
        x, tg, tw = _batch(rank)
        for _ in range(2):
            loss = tr.train_step(x.cuda(), tg.cuda(), tw.cuda(), world_size=world)
        torch.cuda.synchronize()
        q.put((rank, tr.flat[:tr._n_param].cpu().numpy(), float(loss.cpu()[0])))
    finally:
        dist.destroy_process_group()


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
