#!/usr/bin/env python3
import csv, collections, re, sys, glob, os
d = sys.argv[1]
for sub in sorted(glob.glob(os.path.join(d, '*/t_counter_collection.csv'))):
    rows = list(csv.DictReader(open(sub)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in rows:
        m = re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen[k]:
            seen[k].add(r['Dispatch_Id']); dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for k in sorted(agg):
        if 'trace' in k or 'shade' in k:
            n = len(seen[k]); print(os.path.basename(os.path.dirname(sub)), k, n, 'avg_us %.1f' % (dur[k] / n / 1e3), {c: '%.3g' % (v / n) for c, v in agg[k].items()})
