#!/bin/bash
# kernel-trace of tools/attn_bench.py: per-kernel durations of the divided attention launches.  bash tools/attn_trace.sh <tag>
TAG=${1:-attntrace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -f csv -d $O/t -o p -- python $R/tools/attn_bench.py > $O/t.log 2>&1
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob('$O/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if 'egv' in k:
        v2 = sorted(v)
        print(f"{k:72s} n={len(v):4d} med={v2[len(v2)//2]:8.1f} min={v2[0]:8.1f} us")
PY
find $O -name "*.csv" -size +1M -delete
