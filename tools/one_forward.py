#!/usr/bin/env python3
"""One eager HRNet-W32 forward (2N = 128 images) for counter collection under rocprofv3 --pmc."""
import os, sys
if "--per-op" in sys.argv:          # one launch per op (no merged launches): what tools/pmc_by_op.py maps onto the program
    sys.argv.remove("--per-op")
    os.environ["UDP_POSE_NO_GROUPS"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
_, net = bench.build_net(dtype)
net.use_graph = False
hp = bench.HotPath(net, 64, torch.device("cuda", 0), seed=1)
for _ in range(2):
    hp.step()
torch.cuda.synchronize()
