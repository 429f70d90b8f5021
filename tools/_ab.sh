cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --workload sponza_teapots --no-cpu-baseline --no-roofline --sustained-frames 0"
for lib in "" sw4 sw2 "" sw4; do
  if [ -n "$lib" ]; then export TRHIP_LIB=$PWD/tauray_amd/libtrhip_$lib.so; else unset TRHIP_LIB; fi
  echo "== lib=$lib"
  $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['ms_per_frame_sync'])"
done
