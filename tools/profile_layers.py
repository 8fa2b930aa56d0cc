#!/usr/bin/env python3
"""Per-layer-shape timing table of one HRNet-W32 forward (hipEvents around every launch).

    python tools/profile_layers.py [--dtype bf16|f32] [--batch 64] [--reps 5]
"""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    _, net = bench.build_net(a.dtype)
    hp = bench.HotPath(net, a.batch, torch.device("cuda", 0), seed=1)
    hp.step()
    acc = None
    for _ in range(a.reps):
        ms, desc = net.profile(hp.xin, flip_test=True)
        acc = ms if acc is None else acc + ms
    ms = acc / a.reps
    esz = 2 if a.dtype == "bf16" else 4
    b = 2 * a.batch
    rows = OrderedDict()
    for t, (name, kind, ks, st, cin, cout, ho, wo) in zip(ms, desc):
        key = (kind, ks, st, cin, cout, ho, wo)
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += float(t)
    print("%-28s %5s %9s %9s %9s %8s" % ("kind ks st cin cout HxW", "n", "ms_total", "us/launch", "TFLOP/s", "GB/s"))
    tot = 0.0
    for (kind, ks, st, cin, cout, ho, wo), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        fl = 0 if kind == 2 else 2.0 * ks * ks * cin * cout * ho * wo * b * n
        by = (ho * st * wo * st * cin * (4 if kind == 0 else esz) + ho * wo * cout * esz) * b * n
        tot += t
        print("%d %d %d %4d %4d %3dx%-3d        %5d %9.3f %9.1f %9.1f %8.0f" %
              (kind, ks, st, cin, cout, ho, wo, n, t, t / n * 1e3, fl / (t * 1e-3) / 1e12, by / (t * 1e-3) / 1e9))
    print("total kernel ms per forward(2N=%d): %.3f" % (b, tot))


if __name__ == "__main__":
    main()
