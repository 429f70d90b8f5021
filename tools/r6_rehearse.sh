#!/bin/bash
# round 6: the N > 1 bench line rehearsed on one GPU (gloo, all ranks on device 0): the new `ranks` / `display_frame` fields, the default
# flags the driver uses, both exchanges; then __graft_entry__.smoke()
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rehearse; mkdir -p $OUT; cd $R
bash tools/rehearse_8rank.sh 4 auto > $OUT/rehearse_4_auto.txt 2>&1
bash tools/rehearse_8rank.sh 3 ipc > $OUT/rehearse_3_ipc.txt 2>&1
timeout 900 python bench.py --gpus 2 --dist-backend gloo --one-device --steps 20 --warmup 3 > $OUT/n2_default_flags.json 2> $OUT/n2_default_flags.err || echo "2-rank default-flag run failed" >> $OUT/rehearse_4_auto.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
