#!/bin/bash
# schedule sweeps (round 6: profiles/r6/lone_frame_phase_ab.txt, strip_schedule_ab.txt); usage through gpurun: bash tools/schedule_sweep.sh.  (a) a lone frame: enqueue order / lane skew, and more lanes than pipes (libtrhip_lanes8.so);
# (b) a 1/8 strip one frame at a time: lanes, enqueue order
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s6; mkdir -p $OUT; cd $R
export GPU_MAX_HW_QUEUES=8
one() {  # env-set workload -> "sync ms, two-in-flight ms, pipelined ms"
  env $1 python bench.py --workload $2 --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline --sustained-frames 0 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith(chr(123))]
if not l: print('%-70s FAILED' % sys.argv[1]); sys.exit(0)
d=json.loads(l[-1]); print('%-16s %-70s sync %.4f  two %.4f  pipelined %.4f  lanes %s pipes %s' % (sys.argv[2], sys.argv[1], d['ms_per_step'], d['two_in_flight']['ms_per_frame'], d['pipelined']['ms_per_frame'], d['config']['lanes'], d['config']['lane_pipe_classes']))" "$1" $2
}
for rep in 1 2; do
  for wl in sponza_teapots sponza_class; do
    for e in "A=0" "TRHIP_ENQUEUE=lanes" "TRHIP_ENQUEUE=step" "TRHIP_ENQUEUE=skew1" "TRHIP_ENQUEUE=skew2" "TRHIP_ENQUEUE=skew3" \
             "TRHIP_LIB=$R/tauray_amd/libtrhip_lanes8.so TRHIP_LANES=4" "TRHIP_LIB=$R/tauray_amd/libtrhip_lanes8.so TRHIP_LANES=5" \
             "TRHIP_LIB=$R/tauray_amd/libtrhip_lanes8.so TRHIP_LANES=6" "TRHIP_LIB=$R/tauray_amd/libtrhip_lanes8.so TRHIP_LANES=8"; do
      one "$e" $wl
    done
  done
done > $OUT/lone_frame_phase_ab.txt 2>&1
for rep in 1 2; do
  for e in "A=0" "TRHIP_LANES=2" "TRHIP_LANES=3" "TRHIP_ENQUEUE=lanes" "TRHIP_ENQUEUE=skew1" "TRHIP_ENQUEUE=skew2" "TRHIP_MERGED_RAYGEN=0" "TRHIP_SHADE_BLOCKS=1024" "TRHIP_SHADE_BLOCKS=512" \
           "TRHIP_LIB=$R/tauray_amd/libtrhip_lanes8.so TRHIP_LANES=6" "TRHIP_LIB=$R/tauray_amd/libtrhip_lanes8.so TRHIP_LANES=8"; do
    for world in 8 16; do
      env $e python tools/strip_timeline.py render sponza_teapots $world 200 2>&1 | grep "ms per frame" | sed "s|^|$e  |"
    done
  done
done > $OUT/strip_schedule_ab.txt 2>&1
