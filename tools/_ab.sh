cd $GRAFT_REPO_ROOT
for w in sponza_teapots test_glb sponza_class; do
python bench.py --steps 40 --warmup 5 --workload $w --no-cpu-baseline --no-roofline --sustained-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], d['value'], d['ms_per_frame_sync'], d['value_sync_per_frame'])"
done
