# Rehearsal of the N > 1 bench path on a one-GPU box: all ranks on device 0, gloo instead of RCCL.  usage: bash tools/rehearse_multi_rank.sh [F ...]
cd $GRAFT_REPO_ROOT
export TRHIP_BENCH_WATCHDOG=60
for f in ${@:-1 3}; do
timeout 120 python bench.py --steps 5 --frames-in-flight $f --no-cpu-baseline --no-roofline --save-display gpurun_out/disp1.npy > /dev/null
for n in 2 3; do
echo "== F=$f N=$n"
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $n --steps 5 --warmup 2 --frames-in-flight $f --dist-backend gloo --one-device --save-display gpurun_out/disp$n.npy > gpurun_out/rehearse_f${f}_n$n.log 2>&1; grep -n "File \"/root/repo\|File \".*repo\|Error\|error" gpurun_out/rehearse_f${f}_n$n.log | head -30
python - <<PY
import numpy as np
a=np.load('gpurun_out/disp1.npy')
try:
    b=np.load('gpurun_out/disp$n.npy'); print("N=$n equal:", np.array_equal(a,b), float(np.abs(a-b).max()))
except Exception as e: print('missing', e)
PY
done; done
