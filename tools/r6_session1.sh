#!/bin/bash
# round 6, session 1: baseline of the round-5 kernels on this box + the shade phase timeline (full frame and 1/8 strip)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s1; mkdir -p $OUT; cd $R
export GPU_MAX_HW_QUEUES=8
python bench.py --no-pmc --no-cpu-baseline --sustained-frames 0 > $OUT/bench_base.json 2> $OUT/bench_base.err
python tools/shard_share_probe.py sponza_teapots > $OUT/share_base.txt 2> $OUT/share_base.err
for w in 1 8; do
  TRHIP_LIB=$R/tauray_amd/libtrhip_shadetl.so python tools/shade_timeline.py sponza_teapots $w 8 > $OUT/shade_tl_w$w.txt 2> $OUT/shade_tl_w$w.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace8 -o t -- python $R/tools/strip_timeline.py render sponza_teapots 8 40 > $OUT/trace8.log 2>&1
python $R/tools/strip_timeline.py report $(find $OUT/trace8 -name 't_kernel_trace.csv' | head -1) > $OUT/strip_timeline_1_8_base.txt 2>&1
rm -rf $OUT/trace8
