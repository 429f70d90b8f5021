#!/bin/bash
# usage (through gpurun): bash tools/sync_sweep.sh     -> gpurun_out/sync_sweep.txt
# One frame at a time (bench.py `value`) under the schedule knobs, full frame and the strip of a rank of eight: lanes, trace grid, enqueue order,
# hardware queues.  Same bits under every knob (tests/test_gpu_parity.py switch test); this is timing only.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/sync_sweep.txt; : > $OUT
cd $R
run() {   # label, height, env...
  local label=$1 h=$2; shift 2
  local line=$(env "$@" python bench.py --height $h --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline --sustained-frames 0 2>/dev/null | tail -1)
  python - "$label" "$h" "$line" >> $OUT <<'PY'
import json, sys
try:
    r = json.loads(sys.argv[3])
    print("%-44s h=%-5s sync %.4f ms (p50 %.4f) %8.1f Mray/s | pipelined %.4f ms" % (sys.argv[1], sys.argv[2], r["ms_per_step"], r["frame_ms"]["p50"], r["value"], r["pipelined"]["ms_per_frame"]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "failed:", e)
PY
}
for h in 1080 136; do
  run "default" $h TRHIP_X=0
  run "default again" $h TRHIP_X=0
  for l in 1 2 3 4; do run "lanes $l" $h TRHIP_LANES=$l TRHIP_LANES_MIN_PATHS=0; done
  for g in 512 768 1024 1536 2048; do run "trace grid $g" $h TRHIP_GRID_BLOCKS=$g; done
  for s in 512 1024 4096; do run "shade grid $s" $h TRHIP_SHADE_BLOCKS=$s; done
  run "enqueue step" $h TRHIP_ENQUEUE=step
  run "enqueue lanes" $h TRHIP_ENQUEUE=lanes
  run "enqueue skew2" $h TRHIP_ENQUEUE=skew2
  run "4 hardware queues" $h GPU_MAX_HW_QUEUES=4
  run "16 hardware queues" $h GPU_MAX_HW_QUEUES=16
  run "unfused" $h TRHIP_FUSED=0
done
cat $OUT
