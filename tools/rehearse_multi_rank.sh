# Rehearsal of the N > 1 bench path on a one-GPU box: all ranks on device 0, gloo instead of RCCL, ranks started by bench.py itself.  usage: bash tools/rehearse_multi_rank.sh [F ...]
cd $GRAFT_REPO_ROOT
export TRHIP_BENCH_WATCHDOG=60
for f in ${@:-1 4}; do
timeout 180 python bench.py --steps 5 --frames-in-flight $f --no-cpu-baseline --no-roofline --save-display /tmp/disp1.npy > /dev/null
for n in 2 3; do
echo "== F=$f N=$n"
timeout 180 python bench.py --gpus $n --steps 6 --warmup 2 --frames-in-flight $f --dist-backend gloo --one-device --prewarm 200 --save-display /tmp/disp$n.npy > gpurun_out/rehearse_f${f}_n$n.log 2>&1; grep -n "File \"/root/repo\|File \".*repo\|Error\|error" gpurun_out/rehearse_f${f}_n$n.log | head -30
python - <<PY
import numpy as np
a=np.load('/tmp/disp1.npy')
try:
    b=np.load('/tmp/disp$n.npy'); print("N=$n equal:", np.array_equal(a,b), float(np.abs(a-b).max()))
except Exception as e: print('missing', e)
PY
done; done
# the reference's other pixel strategy (interleaved scanlines, no balancing), and strips with equal shares
for extra in "--strategy scanline" "--no-balance"; do
timeout 180 python bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --one-device --prewarm 0 $extra --save-display /tmp/dispx.npy > gpurun_out/rehearse_extra.log 2>&1
python - <<PY
import numpy as np
try:
    print("$extra equal:", np.array_equal(np.load('/tmp/disp1.npy'), np.load('/tmp/dispx.npy')))
except Exception as e: print('$extra missing', e)
PY
done
# view and sample shards (smaller frames: gloo moves device memory through the host)
S="--width 640 --height 360 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline"
T="python bench.py --gpus 2 --dist-backend gloo --one-device --prewarm 0"      # bench.py starts its own ranks
timeout 120 python bench.py $S --views 6 --save-display /tmp/v1.npy > /dev/null
timeout 120 $T $S --views 6 --shard views --save-display /tmp/v2.npy > gpurun_out/rehearse_views.log 2>&1
timeout 120 python bench.py $S --spp 4 --save-display /tmp/s1.npy > /dev/null
timeout 120 $T $S --spp 4 --shard samples --save-display /tmp/s2.npy > gpurun_out/rehearse_samples.log 2>&1
python - <<PY
import numpy as np
try:
    a, b = np.load('/tmp/v1.npy'), np.load('/tmp/v2.npy'); print("views: rank 0 holds", b.shape[0], "of", a.shape[0], "equal:", np.array_equal(a[0::2], b))
except Exception as e: print("views missing", e)
try:
    a, b = np.load('/tmp/s1.npy'), np.load('/tmp/s2.npy'); print("samples: max abs difference of the tonemapped frames", float(np.abs(a - b).max()))
except Exception as e: print("samples missing", e)
PY
