#!/bin/bash
# usage (through gpurun): bash tools/fuzz_builder_switches.sh [combos] [seed]
# Random combinations of the acceleration-structure build switches (builder, PLOC radius, reinsertion rounds, collapse, node layout)
# under the triangle-soup hit-parity test and the full-size property test: hits must stay bit-equal to the oracle's whatever tree is
# built.  (The pre-splitting budget and the treetop rode here until round 4: experiments/r4_retired_switches.patch.)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fuzz; mkdir -p $OUT; cd $R
python - ${1:-12} ${2:-1} > $OUT/combos.txt <<'PY'
import sys, random
random.seed(int(sys.argv[2]))
for _ in range(int(sys.argv[1])):
    e = {"TRHIP_BUILDER": random.choice(["lbvh", "ploc", "ploc"]), "TRHIP_PLOC_RADIUS": random.choice(["4", "16", "64"]), "TRHIP_BVH_OPT": random.choice(["0", "2", "8"]),
         "TRHIP_COLLAPSE": random.choice(["greedy", "cost"]), "TRHIP_NODE_LAYOUT": random.choice(["build", "dfs"])}
    print(" ".join(f"{k}={v}" for k, v in e.items()))
PY
while read -r combo; do
  r=$(env $combo TRHIP_FUZZ_SOUPS="71 72" timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_triangle_soups or full_size_properties" 2>&1 | tail -1)
  echo "$combo -> $r"
done < $OUT/combos.txt
