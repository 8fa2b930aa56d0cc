#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:64]][r["Counter_Name"]] += float(r["Counter_Value"])
key = sys.argv[2] if len(sys.argv) > 2 else "SQ_WAVE_CYCLES"
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0))[:int(sys.argv[3]) if len(sys.argv) > 3 else 6]:
    print(k)
    print("    " + "  ".join("%s=%.3g" % (a.replace("SQ_", ""), b) for a, b in sorted(v.items())))
