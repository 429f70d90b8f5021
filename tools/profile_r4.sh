#!/bin/bash
# usage (through gpurun): bash tools/profile_r4.sh     -> gpurun_out/r4_summary/ (copy into profiles/r4/)
# (a) per workload: rocprofv3 --kernel-trace --stats of the bench command without its own counter passes, and the bench command as the
#     driver runs it with --pmc-dump (as tools/profile_r3.sh);
# (b) the option sets north_star and the reference's presets name, on sponza_teapots: each Sobol sampler and each preset through the program
#     compiled for it, and the A/B legs - the same set through the general kernels at the default arithmetic and at IEEE fp32 (what round 3
#     rendered every non-default set with);
# (c) a kernel trace of one-frame-at-a-time frames for tools/sync_frame_timeline.py.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4_summary; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in ${@:-sponza_teapots test_glb sponza_class}; do
  D=/tmp/prof_$W; rm -rf $D
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- \
      python $R/bench.py --steps 20 --warmup 5 --workload $W --no-pmc --no-cpu-baseline > $OUT/${W}_bench_under_rocprof.json 2> $OUT/${W}_stats.log || echo "stats pass failed ($W)"
  S=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${W}_kernel_stats.csv
  if [ "$W" = sponza_teapots ]; then
    T=$(find $D -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python $R/tools/sync_frame_timeline.py $T > $OUT/sync_frame_timeline.txt 2>&1
  fi
  rm -rf $D
  (cd $R && timeout 600 python bench.py --steps 20 --warmup 5 --workload $W --pmc-dump $OUT > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err) || echo "bench failed ($W)"
done
cd $R
for a in "--sampler 1" "--sampler 2" "--sampler 3" "--preset quality" "--preset reference" "--preset accumulation" \
         "--sampler 1 --general-kernels" "--sampler 1 --general-kernels --ieee-shading" "--preset quality --general-kernels" "--preset quality --general-kernels --ieee-shading" \
         "--general-kernels" "--ieee-shading"; do
  n=$(echo $a | sed 's/--//g; s/ /_/g')
  timeout 300 python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --sustained-frames 0 > $OUT/option_set_$n.json 2> $OUT/option_set_$n.err || echo "failed: $a"
done
timeout 300 python bench.py --sampler 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc-dump $OUT/sobol_owen > $OUT/option_set_sampler_1_with_counters.json 2> $OUT/option_set_sampler_1_with_counters.err
python - $OUT <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        l = [x for x in open(f) if x.startswith("{")]
        r = json.loads(l[-1])
        if "value" not in r: continue
        k = r.get("roofline") or {}
        rows.append("%-52s sync %.4f ms %8.2f Mray/s | pipelined %.4f ms %8.2f | %s | %s" % (os.path.basename(f)[:-5], r["ms_per_step"], r["value"], r["pipelined"]["ms_per_frame"],
                    r["value_pipelined"], r["config"].get("shading_program", ""), ("%s %.3f" % (k.get("bound"), k.get("frac"))) if k.get("frac") else ""))
    except Exception as e:
        rows.append(f"{os.path.basename(f)}: {e}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
