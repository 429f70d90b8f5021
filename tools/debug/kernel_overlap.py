"""Concurrency of the kernels in a rocprofv3 kernel trace (t_kernel_trace.csv of `--kernel-trace`): for the timed region
of a bench run (the stretch where k_trace_fused runs, i.e. profiling off), the wall-clock span, the sum of kernel
durations and how many kernels were running on average and at most.  usage: python tools/debug/kernel_overlap.py <csv>"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if "k_trace_fused" in r["Kernel_Name"]]
lo, hi = fused[0], fused[-1]
# the latency loop (host sync per frame) follows the timed region: keep the first half of the fused stretch
sel = rows[lo:lo + (hi - lo) // 2]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
events = []
busy = 0
per = {}
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    events += [(s, 1), (e, -1)]
    busy += e - s
    k = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"]).group(1)
    per[k] = per.get(k, 0) + (e - s)
events.sort()
cur = peak = 0
covered = 0
last = t0
for t, d in events:
    if cur > 0:
        covered += t - last
    cur += d
    peak = max(peak, cur)
    last = t
span = t1 - t0
print(f"kernels {len(sel)}, span {span / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms, "
      f"average kernels in flight {busy / span:.2f}, peak {peak}, device idle {100 * (1 - covered / span):.1f} % of the span")
print("share of kernel time:", ", ".join(f"{k} {100 * v / busy:.0f} %" for k, v in sorted(per.items(), key=lambda kv: -kv[1])))
