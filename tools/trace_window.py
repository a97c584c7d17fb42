"""Print the kernels of a rocprofv3 --kernel-trace (rocpd sqlite) around the N-th launch of a kernel whose name contains PAT:
stream, start (us, relative), duration, grid, name.   usage: trace_window.py results.db PAT N before_us after_us"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat, nth, before, after = sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
cols = [c[1] for c in db.execute("pragma table_info('kernels')")]
namecol = 'name' if 'name' in cols else 'kernel_name'
gcol = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
wcol = 'workgroup_x' if 'workgroup_x' in cols else ('workgroup_size_x' if 'workgroup_size_x' in cols else None)
sel = f"select {namecol}, start, end, stream_id" + (f", {gcol}" if gcol else ", 0") + (f", {wcol}" if wcol else ", 1") + " from kernels order by start"
rows = db.execute(sel).fetchall()
hits = [r for r in rows if pat in r[0]]
if not hits:
    raise SystemExit(f"no kernel matching {pat}; columns {cols}")
t0 = hits[min(nth, len(hits) - 1)][1]
for n, s, e, sid, g, w in rows:
    if e >= t0 - before * 1e3 and s <= t0 + after * 1e3:
        nm = re.sub(r'^void ', '', n)
        nm = re.sub(r'egv::', '', nm)
        print(f"s{sid:<2d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  wg={(g // w) if w else g:<6d} {nm[:90]}")
