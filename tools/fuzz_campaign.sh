#!/bin/bash
# usage (through gpurun): bash tools/fuzz_campaign.sh "<seeds>" [draws] [small_draws]
# The seeded differential tests of tests/test_gpu_parity.py (option combinations, cameras, light rigs, materials: HIP frame against the
# oracle's) with other seeds and many more draws than the suite runs: a campaign by hand, its tail kept in profiles/<round>/.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fuzz; mkdir -p $OUT
cd $R
for seed in ${1:-101 202 303}; do
  TRHIP_FUZZ_SEED=$seed TRHIP_FUZZ_DRAWS=${2:-300} TRHIP_FUZZ_DRAWS_SMALL=${3:-60} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q \
      -k "random_option_combinations or random_cameras or random_lights or random_materials or random_shard_geometries or random_in_process_jobs or refit_sequences_equal_rebuilds or random_direct_and_gbuffer_targets" > $OUT/seed_$seed.txt 2>&1
  echo "seed $seed: $(tail -1 $OUT/seed_$seed.txt)"
  grep -E "^(FAILED|E  )" $OUT/seed_$seed.txt | head -20
done
# hit parity (bit-exact) on random triangle soups with further seeds
TRHIP_FUZZ_SOUPS="${4:-21 22 23 24 25 26 27 28}" timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_triangle_soups" > $OUT/soups.txt 2>&1
echo "soups: $(tail -1 $OUT/soups.txt)"
grep -E "^(FAILED|E  )" $OUT/soups.txt | head -20
# the C++ host with fake devices
for seed in ${1:-101 202 303}; do
  TRHIP_FUZZ_SEED=$seed TRHIP_FUZZ_DRAWS_SMALL=${3:-60} timeout 1500 python -m pytest tests/test_cpp_host.py -m gpu -q -k cpp_random_multi_device > $OUT/cpp_$seed.txt 2>&1
  echo "cpp host, seed $seed: $(tail -1 $OUT/cpp_$seed.txt)"
  grep -E "^(FAILED|E  )" $OUT/cpp_$seed.txt | cut -c1-400 | head -6
done
