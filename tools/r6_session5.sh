#!/bin/bash
# round 6, session 5: full GPU suite on the tree (light-record fetch, merged raygen, stream pool contract, device / comm info), then the
# shade phase timeline of the adopted kernels with the light sample split up
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s5; mkdir -p $OUT; cd $R
export GPU_MAX_HW_QUEUES=8
python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1
for w in 1 8; do
  TRHIP_LIB=$R/tauray_amd/libtrhip_shadetl.so python tools/shade_timeline.py sponza_teapots $w 8 > $OUT/shade_tl_w$w.txt 2> $OUT/shade_tl_w$w.err
done
python bench.py > $OUT/bench.json 2> $OUT/bench.err
