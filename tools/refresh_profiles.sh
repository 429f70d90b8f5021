#!/bin/bash
# usage (through gpurun): bash tools/refresh_profiles.sh <round>   e.g. r2
# Profiles the bench workloads (tools/profile_round.sh), condenses the raw tables on the box (they are too large to travel back),
# and runs the plain bench lines right after.  Everything lands in gpurun_out/<round>_summary/: copy it into profiles/<round>/.
R=$GRAFT_REPO_ROOT; TAG=${1:-r2}
cd $R
bash tools/profile_round.sh $TAG sponza_teapots test_glb sponza_class > gpurun_out/profile_round.log 2>&1
python tools/profile_summary.py gpurun_out/prof_$TAG gpurun_out/${TAG}_summary
rm -rf gpurun_out/prof_$TAG
for w in sponza_teapots test_glb sponza_class; do
  python bench.py --workload $w > gpurun_out/${TAG}_summary/${w}_bench.json 2> gpurun_out/${TAG}_summary/${w}_bench.err || echo "bench $w failed"
  tail -1 gpurun_out/${TAG}_summary/${w}_bench.json | cut -c1-400
done
ls -la gpurun_out/${TAG}_summary
