cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in test_glb sponza_teapots; do
  O=$R/gpurun_out/tcc_$W; mkdir -p $O
  B="env TRHIP_LANES=1 TRHIP_FUSED=0 python $R/bench.py --steps 4 --warmup 1 --workload $W --no-cpu-baseline --no-roofline"
  timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -o t -- $B > $O/tcc.log 2>&1 || echo fail
  timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/tcp -o t -- $B > $O/tcp.log 2>&1 || echo fail
  python $R/tools/pmc_summary.py $O
done
