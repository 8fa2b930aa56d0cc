#!/usr/bin/env python3
"""Average kernel durations from a rocprofv3 --kernel-trace CSV: python tools/ktrace_summary.py <dir> [substr] [skip_first_n]"""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sub in k:
        rows.setdefault((k, r.get("Grid_Size", "")), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in rows.items():
    v = v[len(v) // 4:]                      # drop warm-up launches
    v.sort()
    print("%-90s grid=%-8s n=%-4d med %7.1f us  min %7.1f" % (k[:90], g, len(v), v[len(v) // 2], v[0]))
