#!/usr/bin/env python3
"""Per-op-shape timing of one RSN-18 forward (config 5): python tools/rsn_layers.py [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import OrderedDict
import torch
import bench
dt = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
_, net = bench.build_net(dt, target_type="offset", model="rsn18")
hp = bench.HotPath(net, 64, torch.device("cuda", 0), seed=1, target_type="offset")
hp.step()
ms, desc = net.profile(hp.xin, flip_test=True)
rows = OrderedDict()
for t, (name, kind, ks, st, cin, cout, ho, wo) in zip(ms, desc):
    r = rows.setdefault((kind, ks, st, cin, cout, ho, wo), [0, 0.0]); r[0] += 1; r[1] += float(t)
for (kind, ks, st, cin, cout, ho, wo), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:14]:
    print("kind %d k%d s%d %4d->%4d %3dx%-3d n=%3d %8.3f ms %8.1f us" % (kind, ks, st, cin, cout, ho, wo, n, t, t / n * 1e3))
print("total %.3f ms over %d ops" % (ms.sum(), len(ms)))
