#!/usr/bin/env python3
"""Print the top_kernels view of a rocprofv3 rocpd .db (name, calls, total us, avg us, %)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = 0.0
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    tot += total
print("total kernel time %.1f us" % tot)
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit %d" % n):
    print("%-72s %6d %10.1f %8.2f %5.1f%%" % (name[:72], calls, total, avg, pct))
