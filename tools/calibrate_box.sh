#!/bin/bash
# usage (through gpurun): bash tools/calibrate_box.sh [out-dir-under-gpurun_out]
# Runs tools/ubench/calibrate plainly (known work + HIP-event times) and under the counter passes bench.py uses, each pass its
# own rocprofv3 run with the kernel trace only.  tools/calibration_summary.py turns the result into profiles/<round>/calibration.json.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-cal}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
grep -o -E "(TCC_EA0?_RDREQ[A-Za-z0-9_]*|TCC_EA0?_WRREQ[A-Za-z0-9_]*|TCP_TCC_[A-Z_]*REQ[A-Za-z0-9_]*|TCC_REQ[A-Za-z0-9_]*|TCC_READ[A-Za-z0-9_]*|TCC_BUBBLE[A-Za-z0-9_]*|TCC_[A-Z_]*DRAM[A-Za-z0-9_]*|TCC_[A-Z0-9_]*MALL[A-Za-z0-9_]*|SQ_INSTS_VALU[A-Za-z0-9_]*|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES|SQ_INST_CYCLES_[A-Z]*)" $OUT/counters_avail.txt | sort -u > $OUT/counters_of_interest.txt
C=$R/tools/ubench/calibrate
$C 3 > $OUT/plain.jsonl 2> $OUT/plain.err
run() { name=$1; shift; timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o t -- $C 1 > $OUT/$name.jsonl 2> $OUT/$name.log || echo "pass $name failed"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum
run hit TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
for extra in TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_RW_READ_REQ_sum; do
  if grep -q "${extra%_sum}" $OUT/counters_avail.txt; then run x_$extra $extra; fi
done
# keep what travels back small: the per-dispatch counter tables only
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
