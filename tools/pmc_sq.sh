#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$2; mkdir -p $O
B="python $R/bench.py --steps 4 --warmup 1 --workload $1 --no-cpu-baseline --no-roofline"
run() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o t -- $B > $O/$name.log 2>&1 || echo "pass $name failed"; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
