#!/bin/bash
# usage (through gpurun): AB_LIBS="libtrhip_head.so libtrhip_x.so ..." bash tools/ab_libs.sh <tag> [workloads]
# Per library: bench.py (lone frame, pipelined, per-kernel ms) three rounds, and the 1/8 strip one frame at a time.
R=$GRAFT_REPO_ROOT; TAG=${1:-ab}; WL=${2:-sponza_teapots}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export GPU_MAX_HW_QUEUES=8
for rep in 1 2 3; do
 for w in $WL; do
  for lib in ${AB_LIBS:-libtrhip_head.so libtrhip.so}; do
    TRHIP_LIB=$R/tauray_amd/$lib python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --sustained-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); k=d['roofline']['kernel_ms_per_frame']; print('%-16s %-24s sync %.4f  two %.4f pipelined %.4f  shade %.4f closest %.4f shadow %.4f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['two_in_flight']['ms_per_frame'], d['pipelined']['ms_per_frame'], k['shade'], k['trace_closest'], k['trace_shadow']))" $w $lib
  done
 done
done > $OUT/bench_ab.txt 2>&1
for lib in ${AB_LIBS:-libtrhip_head.so libtrhip.so}; do
  for rep in 1 2; do
    TRHIP_LIB=$R/tauray_amd/$lib python tools/strip_timeline.py render sponza_teapots 8 200 2>&1 | grep "ms per frame" | sed "s/^/$lib  /"
  done
done > $OUT/strip_ab.txt 2>&1
