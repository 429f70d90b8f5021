#!/bin/bash
# usage (through gpurun): bash tools/profile_r6.sh     -> gpurun_out/r6_summary/ (copy into profiles/r6/)
# (a) per workload: rocprofv3 --kernel-trace --stats of the bench command without its own counter passes, and the bench command as the
#     driver runs it, with --pmc-dump;
# (b) a Sobol sampler and a preset through the programs compiled for them (round 4's table, two rows of it, at this round's kernels);
# (c) the phase timeline of the closest-hit kernel at the final kernels (variant library, tools/trace_timeline.py);
# (d) the lone frame's kernel timeline;
# (e) round 6: the phase timeline of k_shade (variant library libtrhip_shadetl.so, tools/shade_timeline.py) for the whole frame and a 1/8 strip,
#     the strip's kernel timeline (tools/strip_timeline.py) and what a rank's share costs (tools/shard_share_probe.py).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6_summary; mkdir -p $OUT
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
for a in "--sampler 1" "--preset quality"; do
  n=$(echo $a | sed 's/--//g; s/ /_/g')
  timeout 300 python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --sustained-frames 0 > $OUT/option_set_$n.json 2> $OUT/option_set_$n.err || echo "failed: $a"
done
[ -f $R/tauray_amd/libtrhip_timeline.so ] && TRHIP_LIB=$R/tauray_amd/libtrhip_timeline.so timeout 300 python tools/trace_timeline.py sponza_teapots 8 > $OUT/trace_phase_timeline_final_kernels.txt 2> $OUT/timeline.err
if [ -f $R/tauray_amd/libtrhip_shadetl.so ]; then
  TRHIP_LIB=$R/tauray_amd/libtrhip_shadetl.so timeout 300 python tools/shade_timeline.py sponza_teapots 1 8 > $OUT/shade_phase_timeline.txt 2>> $OUT/timeline.err
  TRHIP_LIB=$R/tauray_amd/libtrhip_shadetl.so timeout 300 python tools/shade_timeline.py sponza_teapots 8 8 > $OUT/shade_phase_timeline_strip_1_8.txt 2>> $OUT/timeline.err
fi
timeout 600 python tools/shard_share_probe.py sponza_teapots > $OUT/shard_share_probe_sponza_teapots.txt 2> $OUT/share.err
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace8 -o t -- python $R/tools/strip_timeline.py render sponza_teapots 8 40 > $OUT/trace8.log 2>&1;
 python $R/tools/strip_timeline.py report $(find /tmp/trace8 -name 't_kernel_trace.csv' | head -1) > $OUT/strip_timeline_1_8.txt 2>&1; rm -rf /tmp/trace8)
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
        p = r.get("parity") or {}
        rows.append("%-40s sync %.4f ms %8.2f Mray/s | two in flight %.4f ms %8.2f | pipelined %.4f ms %8.2f | %s | %s | %s" % (os.path.basename(f)[:-5], r["ms_per_step"], r["value"],
                    r["two_in_flight"]["ms_per_frame"], r["value_two_in_flight"], r["pipelined"]["ms_per_frame"], r["value_pipelined"], r["config"].get("shading_program", ""),
                    ("%s %.3f" % (k.get("bound"), k.get("frac"))) if k.get("frac") else "", ("parity outside 1e-2: %.5f" % p["pixels_outside_1e-2"]) if p else ""))
    except Exception as e:
        rows.append(f"{os.path.basename(f)}: {e}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
