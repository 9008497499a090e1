#!/usr/bin/env python3
"""Per-kernel durations of a PacBio bench step from a rocprofv3 results database (kernel trace):
   python tools/ktrace_pb.py <results.db>
Prints totals per kernel and the timeline of the last step; the first plan launch and the last emit launch of a
step run alone on the device (the others share it), so they are the kernels' stand-alone durations."""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
names = {r[0]: r[1] for r in c.execute("select id, kernel_name from %s" % ks)}
rows = list(c.execute("select kernel_id, start, end from %s order by start" % kd))
agg = collections.defaultdict(list)
for k, s, e in rows:
    agg[names[k].split("(")[0][:70]].append((e - s) / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-70s n=%4d total %9.2f ms mean %8.3f min %8.3f max %8.3f" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
pb = [(names[k], s, e) for k, s, e in rows if "pb_" in names[k]]
if pb:
    n = len(pb) // 2
    last = pb[-(n // (len(agg[[k for k in agg if 'pb_plan' in k][0]]) // max(1, n // 8)) if False else 16):]
    t0 = last[0][1]
    for nm, s, e in last:
        print("%-16s %9.2f -> %9.2f  (%7.2f ms)" % ("plan" if "plan" in nm else "emit", (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
