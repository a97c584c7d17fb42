"""Export the kernels of the last full step of a rocprofv3 --kernel-trace result (rocpd sqlite) as a small CSV:
start_us,end_us,stream,grid,wg,name   (a step = the interval between consecutive cast_weights_kernel launches).
usage: trace_dump.py results.db out.csv [steps_back=1]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in db.execute("pragma table_info('kernels')")]
namecol = 'name' if 'name' in cols else 'kernel_name'
gx = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
wx = 'workgroup_x' if 'workgroup_x' in cols else ('workgroup_size_x' if 'workgroup_size_x' in cols else None)
sel = f"select start, end, stream_id, {gx or 0}, {wx or 0}, {namecol} from kernels order by start"
rows = db.execute(sel).fetchall()
marks = [r[0] for r in rows if 'cast_weights_kernel' in r[5]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lo, hi = marks[-1 - back], marks[-1]
with open(sys.argv[2], 'w') as f:
    for s, e, sid, g, w, n in rows:
        if lo <= s < hi:
            n = re.sub(r'^void ', '', n)
            n = re.sub(r'\(.*$', '', n)
            f.write(f"{(s - lo) / 1e3:.2f},{(e - lo) / 1e3:.2f},{sid},{g},{w},{n[:110]}\n")
print("cols:", cols)
