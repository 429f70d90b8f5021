#!/bin/bash
# usage (through gpurun): bash tools/cpp_host_rate.sh [frames]
# Frame times of the C++ host (tauray_hip over include/tauray_hip.hh) on the bench scene, next to bench.py's: one frame at a time
# (-t prints the host wall time of every frame) and with four frame slots (wall time of the run / frames).  No files are written.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/cpp_rate; mkdir -p $OUT; N=${1:-200}
cd $R
python - <<PY
from tauray_amd import scenes
from tauray_amd.scene_io import write_scene_dump
write_scene_dump(scenes.sponza_teapots(width=1920, height=1080), "$OUT/sponza_teapots.trsc")
PY
C="$R/tauray_amd/tauray_hip $OUT/sponza_teapots.trsc --width=1920 --height=1080 --max-ray-depth=4 --filetype=none --skip-nan-check --warmup-frames=20 --frames=$N --headless=$OUT/x"
$C -t > $OUT/sync.txt 2>&1
python - $OUT/sync.txt <<'PY'
import re, sys, numpy as np
h = [float(x) for x in re.findall(r"HOST: ([0-9.]+) ms", open(sys.argv[1]).read())]
p = [float(x) for x in re.findall(r"path tracing \(1 viewports\)\] ([0-9.]+) ms", open(sys.argv[1]).read())]
print("C++ host, one frame at a time: frames", len(h), "host ms mean", round(float(np.mean(h)), 4), "p50", round(float(np.median(h)), 4), "| path tracing ms mean", round(float(np.mean(p)), 4) if p else None)
PY
for F in "2" "4" "4 --frames-per-launch=2" "6 --frames-per-launch=2" "4 --frames-per-launch=4"; do
  $C -t --frames-in-flight=$F 2>&1 | grep "^FRAMES" | sed "s/^/C++ host, slots: /"
done
rm -f $OUT/sponza_teapots.trsc
(timeout 300 python bench.py --no-pmc --no-cpu-baseline --sustained-frames 0 | python -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench.py: sync ms', r['ms_per_step'], 'pipelined ms', r['pipelined']['ms_per_frame'])")
