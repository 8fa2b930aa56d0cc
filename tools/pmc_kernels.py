#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters (counter_collection.csv): python tools/pmc_kernels.py <dir> [substr]"""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sub not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k, a in agg.items():
    n = len(cnt[k])
    w = a.get("SQ_WAVES", 0) / n or 1
    print(k[:110], "dispatches=%d waves=%d" % (n, w))
    print("    " + "  ".join("%s/wave=%.0f" % (c.replace("SQ_", ""), v / n / w) for c, v in sorted(a.items()) if c != "SQ_WAVES"))
