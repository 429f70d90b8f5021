#!/bin/bash
# usage (through gpurun): bash tools/ab_two_libs.sh [pytest -k expression]
# tauray_amd/libtrhip_head.so (a build of another tree, copied there by hand) against tauray_amd/libtrhip.so: three rounds of bench.py per
# workload, back to back on one box (TRHIP_LIB selects the library) - after the parity tests named by the expression, on the library of the tree.
R=$GRAFT_REPO_ROOT; cd $R
[ -n "$1" ] && python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-400 | head
for rep in 1 2 3; do
 for w in sponza_teapots sponza_class test_glb; do
  for lib in ${AB_LIBS:-libtrhip_head.so libtrhip.so}; do
    TRHIP_LIB=$R/tauray_amd/$lib python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --sustained-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); k=d['roofline']['kernel_ms_per_frame']; print('%-16s %-18s sync %.4f  pipelined %.4f  shade %.4f closest %.4f shadow %.4f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['pipelined']['ms_per_frame'], k['shade'], k['trace_closest'], k['trace_shadow']))" $w $lib
  done
 done
done
