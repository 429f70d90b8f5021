#!/bin/bash
# usage (through gpurun): bash tools/presplit_sweep.sh "0 15 30 60" "sponza_teapots sponza_class test_glb"
# bench.py (no counter passes, no CPU baseline) per pre-split budget (TRHIP_PRESPLIT = extra references in percent of the triangles)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/presplit; mkdir -p $OUT
for w in ${2:-sponza_teapots sponza_class test_glb}; do
  for p in ${1:-0 15 30 60}; do
    TRHIP_PRESPLIT=$p TRHIP_DEBUG=1 python $R/bench.py --workload $w --no-pmc --no-cpu-baseline --sustained-frames 0 > $OUT/${w}_$p.json 2> $OUT/${w}_$p.err
    python - $OUT/${w}_$p.json $w $p <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], sys.argv[3], "FAILED"); sys.exit(0)
r = json.loads(l[-1]); k = r["roofline"]
print(sys.argv[2], "presplit", sys.argv[3], "| sync ms", r["ms_per_step"], "Mray/s", r["value"], "| pipelined ms", r["pipelined"]["ms_per_frame"], "Mray/s", r["value_pipelined"],
      "| visits/ray", k["node_visits_per_ray"], "tris/ray", k["tri_tests_per_ray"], "| kernel ms/frame", k["kernel_ms_per_frame"], "| build ms", r["accel_build_ms"])
PY
    grep "pre-split" $OUT/${w}_$p.err | head -1
  done
done
