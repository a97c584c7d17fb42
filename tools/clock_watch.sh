#!/bin/bash
# samples the GPU's shader clock and socket power (rocm-smi) every 0.2 s while a command runs:
#   bash tools/clock_watch.sh <tag> <command ...>   ->  gpurun_out/<tag>/clocks.txt + a summary on stdout
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics Package Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $O/clocks.txt &
W=$!
"$@" > $O/cmd.log 2>&1
kill $W
python - "$O/clocks.txt" <<'PY'
import re, sys
sc, pw = [], []
for line in open(sys.argv[1]):
    m = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', line)
    p = re.search(r'Power \(W\): ([\d.]+)', line)
    if m: sc.append(int(m.group(1)))
    if p: pw.append(float(p.group(1)))
def stats(v): 
    v = sorted(v); return (v[0], v[len(v)//2], v[-1]) if v else None
print('samples', len(sc), 'sclk MHz min/median/max', stats(sc), 'power W min/median/max', stats(pw))
PY
tail -3 $O/clocks.txt
