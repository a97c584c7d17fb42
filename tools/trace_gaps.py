"""GPU idle analysis of a rocprofv3 kernel trace (rocpd sqlite): union of kernel intervals vs wall time for the last steps."""
import sqlite3, sys, re
db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = list(c.execute("select start, end, name from kernels order by start"))
t0, t1 = rows[0][0], rows[-1][1]
# steady state: last 60% of the trace
cut = t0 + int(0.4 * (t1 - t0))
rows = [r for r in rows if r[0] >= cut]
wall = rows[-1][1] - rows[0][0]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
gaps = []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e, _ in rows)
print(f"window {wall/1e6:.1f} ms  union-busy {busy/1e6:.1f} ms ({100*busy/wall:.1f}%)  idle {100*(1-busy/wall):.1f}%  sum of kernel time {ksum/1e6:.1f} ms  (overlap factor {ksum/busy:.2f})")
big = sorted(gaps, reverse=True)[:15]
print("largest gaps (us) before kernel:")
for g, n in big:
    print(f"  {g/1e3:8.1f}  {re.sub(r'[(<].*', '', n)[:60]}")
import collections
h = collections.Counter()
for g, n in gaps:
    h[min(int(g / 1e3) // 5 * 5, 100)] += g
print("idle time by gap size bucket (us -> total ms):", {k: round(v / 1e6, 2) for k, v in sorted(h.items())})
