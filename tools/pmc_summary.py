"""Summarise rocprofv3 --pmc results (rocpd sqlite): per kernel, mean of each counter over dispatches (+ mean duration).
usage: pmc_summary.py results.db [kernel-substring]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ''
rows = db.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection group by kernel_name, counter_name, dispatch_id").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for k, c, d, v, du in rows:
    if sub in k:
        agg[k][c].append(v)
        dur[k].append(du)
for k in agg:
    n = max(len(v) for v in agg[k].values())
    print(f"{k[:100]}  dispatches={n} mean_dur_us={sum(dur[k]) / len(dur[k]) / 1e3:.1f}")
    for c, v in sorted(agg[k].items()):
        print(f"   {c:36s} {sum(v) / len(v):16.1f}")
