#!/bin/bash
# usage (through gpurun): bash tools/ab_libs.sh "<lib tags>" "<workloads>" [extra bench args]
# A/B of libtrhip builds (make -C tauray_amd/csrc variant NAME=<tag> EXTRA=...; tag `main` = libtrhip.so) on one box, back to back:
# bench.py without counter passes and CPU baseline; prints sync / pipelined frame times, kernel times and visit counts.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab; mkdir -p $OUT
TAGS=${1:-main}; WL=${2:-sponza_teapots}; shift; shift
for w in $WL; do
  for t in $TAGS; do
    L=$R/tauray_amd/libtrhip_$t.so; [ "$t" = main ] && L=$R/tauray_amd/libtrhip.so
    TRHIP_LIB=$L python $R/bench.py --workload $w --no-pmc --no-cpu-baseline --sustained-frames 0 "$@" > $OUT/${w}_$t.json 2> $OUT/${w}_$t.err
    python - $OUT/${w}_$t.json $w $t <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], sys.argv[3], "FAILED"); sys.exit(0)
r = json.loads(l[-1]); k = r["roofline"]
print(sys.argv[2], sys.argv[3], "| sync ms", r["ms_per_step"], "Mray/s", r["value"], "| pipelined ms", r["pipelined"]["ms_per_frame"], "Mray/s", r["value_pipelined"],
      "| visits/ray", k["node_visits_per_ray"], "tris/ray", k["tri_tests_per_ray"], "| kernel ms/frame", {a: b for a, b in k["kernel_ms_per_frame"].items() if a in ("trace_closest", "trace_shadow", "shade")})
PY
  done
done
