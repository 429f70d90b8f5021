#!/bin/bash
# usage (through gpurun): bash tools/rehearse_8rank.sh [N] [exchange: auto | ipc]
# The full `bench.py --gpus N` command line on a one-GPU box: N ranks on device 0, gloo instead of RCCL (RCCL refuses two ranks on one
# device), ranks started by bench.py itself, load-balancer phase included.  The display frame of the N ranks must equal the
# single-rank frame bit for bit.
R=$GRAFT_REPO_ROOT; N=${1:-8}; EX=${2:-auto}; OUT=$R/gpurun_out/rehearse_$EX; mkdir -p $OUT
cd $R
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-roofline --sustained-frames 0 --save-display $OUT/disp1.npy > $OUT/n1.json 2> $OUT/n1.err || echo "single-rank run failed"
timeout 900 python bench.py --gpus $N --dist-backend gloo --one-device --exchange $EX --steps 20 --no-roofline --save-display $OUT/disp$N.npy > $OUT/n$N.json 2> $OUT/n$N.err || { echo "$N-rank run failed"; tail -30 $OUT/n$N.err; }
python - $OUT $N <<'PY'
import json, sys, numpy as np
out, n = sys.argv[1], sys.argv[2]
a, b = np.load(f"{out}/disp1.npy"), np.load(f"{out}/disp{n}.npy")
print("display frames equal:", bool(np.array_equal(a, b)), a.shape, float(a[..., :3].mean()))
for t in ("1", n):
    l = [x for x in open(f"{out}/n{t}.json") if x.startswith("{")]
    r = json.loads(l[-1])
    print(t, "ranks:", {k: r[k] for k in ("value", "ms_per_step", "value_pipelined", "n_gpus", "steps_effective")}, r["config"]["parallelism"], r["config"].get("exchange"), r.get("load_balance", {}).get("workloads"), r.get("rank_phases"))
PY
rm -f $OUT/disp1.npy $OUT/disp$N.npy      # 33 MB each: gpurun copies at most 64 MiB back
