"""The kernels of a `rocprofv3 --kernel-trace --stats` run by total time.  usage: python tools/kernel_stats_top.py DIR [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -int(r["TotalDurationNs"]))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print("%8.3f ms %5d calls %8.1f us avg  %s" % (int(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total %.3f ms" % (tot / 1e6))
