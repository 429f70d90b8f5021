#!/bin/bash
# usage: tools/pmc_diag.sh <workload> <tag>   (env switches such as TRHIP_LANES / TRHIP_FUSED / TRHIP_BUILDER are inherited) -- run on the GPU box
# every pass is wrapped in `timeout`: a rejected counter set makes rocprofv3 hang after its abort.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$2; mkdir -p $O
B="python $R/bench.py --steps 4 --warmup 1 --workload $1 --no-cpu-baseline --no-roofline"
run() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o t -- $B > $O/$name.log 2>&1 || echo "pass $name failed"; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcc TCC_HIT_sum TCC_MISS_sum
