import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from collections import OrderedDict
_, net = bench.build_net("f16x2", model="w48")
hp = bench.HotPath(net, 32, torch.device("cuda", 0), seed=1, h=384, w=288)
hp.step()
ms, desc = net.profile(hp.xin, flip_test=True)
rows = OrderedDict()
for t, (name, kind, ks, st, cin, cout, ho, wo) in zip(ms, desc):
    r = rows.setdefault((kind, ks, st, cin, cout, ho, wo), [0, 0.0]); r[0] += 1; r[1] += float(t)
b = 64
for (kind, ks, st, cin, cout, ho, wo), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:14]:
    fl = 2.0 * ks * ks * cin * cout * ho * wo * b * n
    by = (ho * st * wo * st * cin + ho * wo * cout) * 4 * b * n
    print("%d k%d s%d %4d->%4d %3dx%-3d n=%3d %8.3f ms %8.1f us %7.1f TF %6.0f GB/s" % (kind, ks, st, cin, cout, ho, wo, n, t, t / n * 1e3, fl / t / 1e9, by / t / 1e6))
print("total", ms.sum())
