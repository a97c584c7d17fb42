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
# GPU busy time = union of all kernel intervals; per HIP stream: sum of its kernels (how the work is spread over the streams)
if 'stream_id' in cols:
    iv = sorted(db.execute(f"select start, end, stream_id from {view}").fetchall())
    busy, cur_s, cur_e = 0, None, None
    per_stream = collections.defaultdict(float)
    for s0, e0, sid in iv:
        per_stream[sid] += (e0 - s0) / 1e6
        if cur_e is None or s0 > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s0, e0
        else:
            cur_e = max(cur_e, e0)
    busy += (cur_e - cur_s) if cur_e is not None else 0
    span = (max(e for _, e, _ in iv) - min(s0 for s0, _, _ in iv)) / 1e6
    # idle gaps of the busiest stream (the critical path of a step): total and the largest ones with their neighbours
    main = max(per_stream, key=per_stream.get)
    ks = sorted(db.execute(f"select start, end, {namecol} from {view} where stream_id = ?", (main,)).fetchall())
    gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1) if ks[i + 1][0] > ks[i][1]]
    tot_gap = sum(g0 for g0, _, _ in gaps if g0 < 50e6)
    big = sorted([g0 for g0 in gaps if g0[0] < 50e6], reverse=True)[:int(6 * steps)]
    print(f"Stream {main}: idle between its kernels {tot_gap / 1e6 / steps:.1f} ms/step; gaps > 100 us: "
          f"{sum(g0 for g0, _, _ in gaps if 100e3 < g0 < 50e6) / 1e6 / steps:.1f} ms/step, 20-100 us: {sum(g0 for g0, _, _ in gaps if 20e3 < g0 <= 100e3) / 1e6 / steps:.1f}, "
          f"< 20 us: {sum(g0 for g0, _, _ in gaps if g0 <= 20e3) / 1e6 / steps:.1f}")
    for g0, a, b in big[:12]:
        print(f"   gap {g0 / 1e3:8.1f} us after `{re.sub(r'[(<].*$', '', a)[:50]}` before `{re.sub(r'[(<].*$', '', b)[:50]}`")
    # steady-state view: a step = the interval between consecutive cast_weights_kernel launches (one per step); the last full
    # ones are the timed steps.  Busy / idle of the calling stream inside them, and the largest gaps with their neighbours.
    marks = [k0 for k0, _, n in ks if 'cast_weights_kernel' in n]
    if len(marks) >= 4:
        lo, hi = marks[-4], marks[-1]
        inside = [(a0, b0, n) for a0, b0, n in ks if lo <= a0 < hi]
        nst = 3
        busy_in = sum(b0 - a0 for a0, b0, _ in inside)
        g_in = [(inside[i + 1][0] - inside[i][1], inside[i][2], inside[i + 1][2]) for i in range(len(inside) - 1) if inside[i + 1][0] > inside[i][1]]
        print(f"Last {nst} steps (between cast_weights launches): {(hi - lo) / 1e6 / nst:.1f} ms/step wall under the profiler, stream {main} busy "
              f"{busy_in / 1e6 / nst:.1f} ms/step, idle {sum(x for x, _, _ in g_in) / 1e6 / nst:.1f} ms/step "
              f"(gaps > 100 us: {sum(x for x, _, _ in g_in if x > 100e3) / 1e6 / nst:.1f}, 10-100 us: {sum(x for x, _, _ in g_in if 10e3 < x <= 100e3) / 1e6 / nst:.1f}, "
              f"< 10 us: {sum(x for x, _, _ in g_in if x <= 10e3) / 1e6 / nst:.1f} over {len(g_in) / nst:.0f} gaps)")
        for x, a_, b_ in sorted(g_in, reverse=True)[:14]:
            print(f"   gap {x / 1e3:8.1f} us after `{re.sub(r'[(<].*$', '', a_)[:48]}` before `{re.sub(r'[(<].*$', '', b_)[:48]}`")
    print()
    print(f"Kernel-interval union (GPU busy) {busy / 1e6 / steps:.1f} ms/step of a {span / steps:.1f} ms/step trace span; per stream (ms/step): "
          + ", ".join(f"stream {k}: {v / steps:.1f}" for k, v in sorted(per_stream.items(), key=lambda kv: -kv[1])[:4]) + "\n")
agg = collections.defaultdict(list)
for n, s, e in rows:
    n = re.sub(r'\(.*$', '', n)
    agg[n].append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"Total kernel time {tot / 1e3:.1f} ms over {steps:g} steps = **{tot / 1e3 / steps:.1f} ms/step**; {sum(len(v) for v in agg.values()) / steps:.0f} launches/step\n")
print("| kernel | calls/step | ms/step | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"| `{n[:100]}` | {len(v) / steps:.1f} | {sum(v) / 1e3 / steps:.2f} | {100 * sum(v) / tot:.1f} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")
