#!/bin/bash
# round 6, session 4: light-record fetch + exact fast texel wrap (+ merged raygen) against the round-5 library and the light-only build;
# strip timeline of the tree's library; the GPU tests touched by the stream pool / device info changes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT; cd $R
export GPU_MAX_HW_QUEUES=8
for mode in 1 0; do
  for lib in libtrhip_head.so libtrhip.so; do
    for wl in sponza_teapots test_glb; do
      TRHIP_SHADE_FAST=$mode TRHIP_LIB=$R/tauray_amd/$lib python tools/ab_frame.py /tmp/frame_${wl}_${lib}_$mode.npy $wl > /dev/null 2>$OUT/ab_frame_${lib}_$mode.err
    done
  done
  for wl in sponza_teapots test_glb; do python tools/ab_frame.py --compare /tmp/frame_${wl}_libtrhip_head.so_$mode.npy /tmp/frame_${wl}_libtrhip.so_$mode.npy >> $OUT/ab_identity.txt 2>&1; done
done
python -m pytest tests -m gpu -q -x -k "stream_pool or abi or specialization or texture" > $OUT/pytest_subset.log 2>&1
AB_LIBS="libtrhip_head.so libtrhip_lightonly.so libtrhip.so" bash tools/r6_ab_libs.sh s4 "sponza_teapots sponza_class test_glb"
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace8_$m -o t -- env TRHIP_MERGED_RAYGEN=$m python $R/tools/strip_timeline.py render sponza_teapots 8 40 > $OUT/trace8_$m.log 2>&1
python $R/tools/strip_timeline.py report $(find $OUT/trace8_$m -name 't_kernel_trace.csv' | head -1) > $OUT/strip_timeline_1_8_merged_raygen_$m.txt 2>&1
rm -rf $OUT/trace8_$m
done
