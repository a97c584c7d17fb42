"""summarise rocprofv3 --pmc counter_collection csv: per (kernel, grid) mean of each counter and mean duration"""
import csv, sys, collections, re, glob
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = (re.sub(r'\(.*$', '', r['Kernel_Name'])[:60], r.get('Grid_Size', ''))
        rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if 'Start_Timestamp' in r:
            rows[k]['dur_us'].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
for k, d in sorted(rows.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, 'n=', max(len(v) for v in d.values()))
