#!/usr/bin/env python3
"""Offline report of a tools/stamp_multi.py dump: python tools/stamp_multi_report.py dump.npy n0,n1,.. [lpt]
n_j = workgroups of member j (deepest first); third argument "lpt" if the dump was taken with UDP_POSE_WS_ORDER=lpt."""
import collections, sys
import numpy as np
F = 2400.0   # s_memtime ticks per us on gfx950 (shader clock while the launch runs)

def order(counts, mode, bias=125, segs_max=64):
    """Member of every flat workgroup: port of ws_order (csrc/conv.hip).  mode "lpt": member after member."""
    n = len(counts); lo = [0] * n; segs = []; total = sum(counts)
    def take(j, c):
        c = min(c, counts[j] - lo[j])
        if c: segs.append((j, lo[j], c)); lo[j] += c
    if mode != "lpt":
        g = 32
        while (total + g - 1) // g > segs_max - 2 * n: g += 8
        nb_, na = counts[-1], total - counts[-1]; ta = tb = 0
        while ta < na and tb < nb_ and len(segs) < segs_max - n:
            if ta * nb_ * 100 <= tb * na * bias:
                j = next(k for k in range(n) if lo[k] < counts[k]); c = min(counts[j] - lo[j], g); take(j, c); ta += c
            else:
                c = min(nb_ - tb, g); take(n - 1, c); tb += c
    for j in range(n): take(j, counts[j])
    mem = []
    for j, f, c in segs: mem += [j] * c
    return np.array(mem), segs

s = np.load(sys.argv[1]).astype(np.int64)
counts = [int(x) for x in sys.argv[2].split(",")]
mem, segs = order(counts, sys.argv[3] if len(sys.argv) > 3 else "")
nwg = len(mem); s = s[:nwg]
hw = s[:, 0, 8] & 0xFFFFFFFF; xcc = (s[:, 0, 8] >> 32) & 0xF
cu = xcc * 4096 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 32 + ((hw >> 8) & 0xF)
start = s[:, :, 0].min(axis=1).astype(np.float64); end = s[:, :, 7].max(axis=1).astype(np.float64)
for x in set(cu.tolist()):      # every CU's counter has its own offset; its first workgroup starts with the launch
    m = cu == x; o = start[m].min(); start[m] -= o; end[m] -= o; s[m, :, :8] -= int(o)
start /= F; end /= F; life = end - start
print("segments:", segs[:12], "..." if len(segs) > 12 else "")
print("launch span %.1f us, %d CUs" % (end.max(), len(set(cu.tolist()))))
nm = len(counts)
for j in range(nm):
    m = mem == j
    print("member %d: %4d wgs  start %.1f..%.1f  end %.1f..%.1f  life mean %.1f (p10 %.1f p90 %.1f)" % (
        j, m.sum(), start[m].min(), start[m].max(), end[m].min(), end[m].max(), life[m].mean(), *np.percentile(life[m], [10, 90])))
d = np.diff(s[:, :, :8].astype(np.float64), axis=2) / F
names = ["prologue", "issue", "waitDMA", "barrier", "MFMAloop", "epi", "drain"]
for j in range(nm):
    m = mem == j
    print("member %d phases us: " % j + "  ".join("%s %.2f" % (n, d[m][:, :, k].mean()) for k, n in enumerate(names)))
per = collections.defaultdict(list)
for i in range(nwg): per[int(cu[i])].append(i)
frac = np.zeros((nwg, nm + 1))
for c, idx in per.items():
    for i in idx:
        for k in idx:
            if k != i:
                frac[i, mem[k]] += max(0.0, min(end[i], end[k]) - max(start[i], start[k])) / life[i]
        frac[i, -1] = max(0.0, 1 - frac[i, :-1].sum())
for j in range(nm):
    m = mem == j
    print("member %d partner shares (m0..,alone):" % j, np.round(frac[m].mean(axis=0), 2), end="")
    for k in range(nm):
        sel = m & (frac[:, k] > 0.7)
        if sel.sum() > 5: print("  | beside m%d: n=%d life %.1f" % (k, sel.sum(), life[sel].mean()), end="")
    print()
T = np.arange(0, end.max(), 4.0)
for j in range(nm):
    m = mem == j
    print("running m%d:" % j, " ".join("%3d" % ((start[m] <= t) & (end[m] > t)).sum() for t in T))
