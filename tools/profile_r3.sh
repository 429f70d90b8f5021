#!/bin/bash
# usage (through gpurun): bash tools/profile_r3.sh [workload ...]     -> gpurun_out/r3_summary/ (copy into profiles/r3/)
# Per workload: (a) rocprofv3 --kernel-trace --stats of the bench command without its own counter passes (a profiler inside a
# profiled process does not nest): the kernel table the roofline's avg_launch_ms must agree with; (b) the bench command as the
# driver runs it (python bench.py --steps 20 --warmup 5), whose roofline object comes from the counter passes it runs itself;
# their per-launch counter averages are kept next to the line (--pmc-dump).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3_summary; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in ${@:-sponza_teapots test_glb sponza_class}; do
  D=/tmp/prof_$W; rm -rf $D
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- \
      python $R/bench.py --steps 20 --warmup 5 --workload $W --no-pmc --no-cpu-baseline > $OUT/${W}_bench_under_rocprof.json 2> $OUT/${W}_stats.log || echo "stats pass failed ($W)"
  S=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${W}_kernel_stats.csv
  rm -rf $D
  (cd $R && timeout 600 python bench.py --steps 20 --warmup 5 --workload $W --pmc-dump $OUT > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err) || echo "bench failed ($W)"
  python - $OUT/${W}_bench.json $OUT/${W}_kernel_stats.csv $W <<'PY'
import csv, json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
r = json.loads(l[-1]); k = r["roofline"]
print(sys.argv[3], "| value", r["value"], "Mray/s,", r["ms_per_step"], "ms sync | pipelined", r["value_pipelined"], r["pipelined"]["ms_per_frame"], "ms | roofline", k["bound"], k["frac"],
      {n: d["frac"] for n, d in k["levels"].items()}, "avg launch", k["avg_launch_ms"], "ms | cpu", r.get("cpu_baseline", {}).get("value"))
try:
    for row in csv.DictReader(open(sys.argv[2])):
        if "k_trace_closest<false, true" in row["Name"]:
            print("   rocprofv3 row:", row["Name"][:60], "calls", row["Calls"], "avg ns", row["AverageNs"])
except Exception as e:
    print("   (no kernel stats:", e, ")")
PY
done
ls -la $OUT
