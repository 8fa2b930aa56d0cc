#!/usr/bin/env python3
"""Config 3 (training step, function.py:38-77) timing on one GPU: W32 256x192, batch 32, Adam.
Prints ms/step and images/s; `--prof` wraps nothing -- run under rocprofv3 --kernel-trace --stats."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from udp_pose_amd import synth
from udp_pose_amd.train import HRNetTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--target", default="gaussian")
ap.add_argument("--backend", default="nccl")
ap.add_argument("--no-graph", action="store_true", help="one ctypes launch per kernel instead of replaying the captured step")
a = ap.parse_args()
# one process per GPU under torch.distributed.run (backend nccl = RCCL); plain `python` = one GPU
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if world > 1:
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if a.backend == "nccl" else 0)
    # --backend gloo: several ranks on ONE GPU (the one-GPU test box); RCCL needs a device per rank
    dist.init_process_group(a.backend, rank=rank, world_size=world)
nj = 17
sd = synth.synth_state_dict(synth.W32_EXTRA, nj, a.target, seed=0)
tr = HRNetTrainer({"MODEL": {"EXTRA": synth.W32_EXTRA, "NUM_JOINTS": nj, "TARGET_TYPE": a.target}}, sd, dtype=a.dtype)
x = torch.from_numpy(synth.synth_crops(a.batch, 256, 192, seed=1 + rank)).cuda()
c = nj * (3 if a.target == "offset" else 1)
tg = torch.from_numpy(synth.synth_heatmaps(a.batch, nj, 64, 48, seed=2, channels_per_joint=c // nj)).cuda()
tw = torch.ones(a.batch, nj, 1, device="cuda")
graphed = not a.no_graph          # N > 1: graph segments between the gradient buckets' all-reduces (train.py)


def step():
    return tr.train_step_graphed(x, tg, tw, world_size=world) if graphed else tr.train_step(x, tg, tw, world_size=world)


for _ in range(max(a.warmup, 2 if graphed else 0)):
    loss = step()
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    loss = step()
e1.record()
torch.cuda.synchronize()
wall = (time.time() - t0) / a.steps * 1e3
dev = e0.elapsed_time(e1) / a.steps
if world > 1:
    t = torch.tensor([wall], device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    dist.barrier()
    dist.destroy_process_group()
if rank == 0:
    import json
    from udp_pose_amd.hrnet_plan import HRNetProgram
    macs = HRNetProgram(sd, synth.W32_EXTRA, 256, 192, "f32").macs_per_image()
    flops = 3 * 2.0 * macs * a.batch                     # forward + input gradient + weight gradient
    peak = {"f32": 157.3, "bf16": 2500.0}[a.dtype]
    tf = flops / (dev * 1e-3) / 1e12
    print(json.dumps({"metric": "images/sec HRNet-W32 256x192 training step (fwd + JointsMSELoss + bwd + Adam"
                                + (" + gradient all-reduce)" if world > 1 else ")"),
                      "value": round(a.batch * world / wall * 1e3, 1), "unit": "images/s", "n_gpus": world,
                      "ms_per_step": round(wall, 3), "dtype": a.dtype, "batch_per_gpu": a.batch, "hipgraph": graphed,
                      "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(tf / peak, 4),
                                   "note": "algorithmic FLOPs of the three conv passes per step / device time of the "
                                           "whole step; the step is bound by its many small BatchNorm / element-wise "
                                           "launches, not by the matrix pipe (profiles/r02_train_*_kernel_stats)"}}))
    print("world %d (global batch %d): %.0f img/s" % (world, a.batch * world, a.batch * world / wall * 1e3))
    print("train W32 b=%d/GPU %s%s: %.1f ms/step (device %.1f ms), %.0f img/s per GPU, loss %s after %d steps, peak mem %.1f GiB" % (
        a.batch, a.dtype, " hipGraph replay" if graphed else "", wall, dev, a.batch / wall * 1e3, loss.cpu().numpy(), tr.step_count,
        torch.cuda.max_memory_allocated() / 2**30))
