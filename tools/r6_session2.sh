#!/bin/bash
# round 6, session 2: the batched k_shade (material block, texture table entries, light records) + merged raygen against the round-5 library
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s2; mkdir -p $OUT; cd $R
export GPU_MAX_HW_QUEUES=8
for mode in 1 0; do
  for lib in libtrhip_head.so libtrhip.so; do
    TRHIP_SHADE_FAST=$mode TRHIP_LIB=$R/tauray_amd/$lib python tools/ab_frame.py /tmp/frame_${lib}_$mode.npy sponza_teapots > /dev/null 2>$OUT/ab_frame_${lib}_$mode.err
  done
  python tools/ab_frame.py --compare /tmp/frame_libtrhip_head.so_$mode.npy /tmp/frame_libtrhip.so_$mode.npy >> $OUT/ab_identity.txt 2>&1
done
python -m pytest tests -m gpu -q -x -k "parity or stream_pool or specialization or abi" > $OUT/pytest_subset.log 2>&1
bash tools/ab_two_libs.sh > $OUT/ab_two_libs.txt 2>&1
python tools/shard_share_probe.py sponza_teapots > $OUT/share_new.txt 2> $OUT/share_new.err
for w in 1 8; do
  TRHIP_LIB=$R/tauray_amd/libtrhip_shadetl.so python tools/shade_timeline.py sponza_teapots $w 8 > $OUT/shade_tl_w$w.txt 2> $OUT/shade_tl_w$w.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace8 -o t -- python $R/tools/strip_timeline.py render sponza_teapots 8 40 > $OUT/trace8.log 2>&1
python $R/tools/strip_timeline.py report $(find $OUT/trace8 -name 't_kernel_trace.csv' | head -1) > $OUT/strip_timeline_1_8.txt 2>&1
rm -rf $OUT/trace8
