#!/bin/bash
# usage (through gpurun): bash tools/ab_env.sh "<VAR=value,... sets separated by spaces; `-` = none>" "<workloads>" [extra bench args]
# A/B of environment switches on one box, back to back, with the tree's libtrhip.so.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab; mkdir -p $OUT
SETS=${1:--}; WL=${2:-sponza_teapots}; shift; shift
for w in $WL; do
  for e in $SETS; do
    envs=""; [ "$e" != "-" ] && envs=$(echo $e | tr ',' ' ')
    tag=$(echo $e | tr -c 'A-Za-z0-9=\n' '_')
    env $envs python $R/bench.py --workload $w --no-pmc --no-cpu-baseline --sustained-frames 0 "$@" > $OUT/${w}_$tag.json 2> $OUT/${w}_$tag.err
    python - $OUT/${w}_$tag.json $w "$e" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], sys.argv[3], "FAILED"); sys.exit(0)
r = json.loads(l[-1]); k = r.get("roofline") or {}
print(sys.argv[2], sys.argv[3], "| sync ms", r["ms_per_step"], "Mray/s", r["value"], "| two in flight ms", (r.get("two_in_flight") or {}).get("ms_per_frame"), "| pipelined ms", r["pipelined"]["ms_per_frame"], "Mray/s", r["value_pipelined"],
      "| kernel ms/frame", {a: b for a, b in (k.get("kernel_ms_per_frame") or {}).items() if a in ("trace_closest", "trace_shadow", "shade")})
PY
  done
done
