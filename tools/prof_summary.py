"""Summarise a rocprofv3 --kernel-trace results.db (rocpd sqlite) into a markdown table under profiles/."""
import re, sqlite3, sys
db, out, title, nsteps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
with open(out, 'w') as f:
    f.write(f"# {title}\n\n")
    f.write("Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-events`\n")
    f.write("(BASELINE.json configs[2]: full fusion EgoNCE+MLM+ITM, B=8, 16x224^2 frames, 32 tokens, bf16 storage, 1 x MI355X;\n")
    f.write(f"the trace holds {nsteps} steps including the warm-up step).\n\n")
    f.write(f"Total kernel time {tot:.1f} ms over {nsteps} steps = **{tot/nsteps:.1f} ms/step**.\n\n")
    f.write("| kernel | calls/step | ms/step | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|\n")
    for n, cnt, ms, avg, mn, mx in rows[:40]:
        n = re.sub(r'\(.*', '', n).replace('void ', '')
        f.write(f"| `{n[:95]}` | {cnt/nsteps:.1f} | {ms/nsteps:.2f} | {100*ms/tot:.1f} | {avg:.1f} | {mn:.1f} | {mx:.1f} |\n")
print(open(out).read()[:1500])
