#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/profile_round.sh <tag> [workload ...]
# Writes gpurun_out/prof_<tag>/<workload>/{stats,fetch,write,sq,ta}/... ; summarise with tools/profile_summary.py and
# copy the summaries into profiles/<round>/.  Counter passes are separate runs (no trace domains besides the kernel
# trace), at most two TA/TCC counters per pass, and every pass sits under `timeout`: a counter set the hardware rejects
# makes rocprofv3 hang after its abort.
TAG=$1; shift
WORKLOADS=${@:-test_glb sponza_teapots}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
  O=$R/gpurun_out/prof_$TAG/$W; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- \
      python $R/bench.py --steps 20 --warmup 3 --workload $W --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log || echo "stats pass failed ($W)"
  # counter passes: one lane and unfused launches, so that a k_trace_closest launch is a whole queue of one bounce like
  # the launch the roofline refers to (the profiler serialises kernels under --pmc anyway)
  B="env TRHIP_LANES=1 TRHIP_FUSED=0 python $R/bench.py --steps 6 --warmup 2 --workload $W --no-cpu-baseline --no-roofline"      # an even number of steps: two frames per launch, like the default bench line
  run() { name=$1; shift; timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o t -- $B > $O/$name.log 2>&1 || echo "pass $name failed ($W)"; }
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
  run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
done
