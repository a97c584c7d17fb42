"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite): per kernel calls/step, ms/step, avg/min/max us.
usage: prof_summary.py results.db steps [top]"""
import sqlite3, sys, collections, re
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = 'kernels' if 'kernels' in tabs else None
if view is None:
    raise SystemExit(f"no kernels view; tables: {tabs[:40]}")
cols = [c[1] for c in db.execute(f"pragma table_info('{view}')")]
namecol = 'name' if 'name' in cols else 'kernel_name'
rows = db.execute(f"select {namecol}, start, end from {view}").fetchall()
agg = collections.defaultdict(list)
for n, s, e in rows:
    n = re.sub(r'\(.*$', '', n)
    agg[n].append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"Total kernel time {tot / 1e3:.1f} ms over {steps:g} steps = **{tot / 1e3 / steps:.1f} ms/step**; {sum(len(v) for v in agg.values()) / steps:.0f} launches/step\n")
print("| kernel | calls/step | ms/step | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"| `{n[:100]}` | {len(v) / steps:.1f} | {sum(v) / 1e3 / steps:.2f} | {100 * sum(v) / tot:.1f} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")
