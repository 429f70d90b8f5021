#!/bin/bash
# the round's last look at the final tree, as the driver will run it: the -m gpu suite, smoke(), the default bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; cd $R
python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
