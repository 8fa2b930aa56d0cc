#!/usr/bin/env python3
"""Experiment: K independent half/quarter batches on K streams (each its own graph replay) vs one batch on
one stream.  Kernels of different streams may co-execute and fill each other's ramp-up / drain gaps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"
MODEL = sys.argv[2] if len(sys.argv) > 2 else "w32"
CASES = [tuple(int(v) for v in a.split(":")) for a in sys.argv[3:]] or [(64, 1), (64, 2), (64, 4), (128, 2)]


def run(total, k, steps=30, warmup=5):
    dev = torch.device("cuda", 0)
    hps, streams = [], []
    for i in range(k):
        _, net = bench.build_net(DT, target_type="offset" if MODEL == "rsn18" else "gaussian", model=MODEL)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            _, h, w, _ = bench.MODELS_CFG[MODEL]
            hps.append(bench.HotPath(net, total // k, dev, seed=1 + i, h=h, w=w, target_type="offset" if MODEL == "rsn18" else "gaussian"))
        streams.append(s)
    def step():
        for hp, s in zip(hps, streams):
            with torch.cuda.stream(s):
                hp.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    print("batch %d as %d stream(s): %.3f ms/step, %.0f img/s" % (total, k, dt * 1e3, total / dt), flush=True)

for total, k in CASES:
    run(total, k)
