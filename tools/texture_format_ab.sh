#!/bin/bash
# usage (through gpurun): bash tools/texture_format_ab.sh     -> gpurun_out/texture_format_ab.txt
# The bench line of several builds of the tree on one box, back to back, three rounds: `.` and whatever of ab_old/, ab_branch/ exists next to it
# (git worktrees of other commits, built with make -C <tree>/tauray_amd/csrc ../libtrhip.so; they travel with the snapshot and are not committed).
# Round 4 used it for profiles/r4/texture_format_ab.txt.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/texture_format_ab.txt; : > $OUT
for round in 1 2 3; do
for t in ab_old ab_branch .; do [ -d $R/$t ] || continue
  for W in sponza_teapots sponza_class; do
  cd $R/$t
  line=$(python bench.py --workload $W --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --sustained-frames 0 2>/dev/null | tail -1)
  python - "$t" "$W" "$line" >> $OUT <<'PY'
import json, sys
r = json.loads(sys.argv[3]); k = r.get("roofline", {}); km = r.get("kernels_serialised_ms_per_frame") or {}
print("%-10s %-15s sync %.4f ms | pipelined %.4f ms | k_trace_closest alone %.4f ms | %s" % (sys.argv[1], sys.argv[2], r["ms_per_step"], r["pipelined"]["ms_per_frame"], k.get("avg_launch_ms", 0), {a: b for a, b in r.items() if "kernel" in a and isinstance(b, dict)}))
PY
  done
done
done
cat $OUT
