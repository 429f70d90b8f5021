#!/usr/bin/env python3
"""Condenses the counter passes of tools/pmc_survey.sh: per kernel family, the average of every counter per launch and the launch time.
usage: python tools/pmc_survey.py gpurun_out/survey_<workload>"""
import collections
import csv
import glob
import json
import os
import re
import sys

src = sys.argv[1]
out = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)<", r["Kernel_Name"])
        if not m or m.group(1) not in ("k_trace_closest", "k_trace_shadow", "k_shade", "k_trace_fused"):
            continue
        key = m.group(1)
        if key == "k_trace_closest" and "<false, true" not in r["Kernel_Name"]:
            continue
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[key]:
            seen[key].add(r["Dispatch_Id"])
            dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in agg:
        n = len(seen[k])
        for c, v in agg[k].items():
            out[k][c] = v / n
            out[k]["us:" + c] = dur[k] / n / 1e3
for k, d in out.items():
    print(k)
    for c in sorted(x for x in d if not x.startswith("us:")):
        print(f"  {c:48s} {d[c]:16.0f}   launch {d['us:' + c]:8.1f} us")
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1, sort_keys=True)
