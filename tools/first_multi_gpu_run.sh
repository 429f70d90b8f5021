#!/bin/bash
# The first lease of a node with two or more MI355X: correctness of every exchange between real devices first
# (tests/test_multi_device.py - skipped on the one-GPU boxes everything else was developed on), then the A/B matrix DESIGN.md section 6
# describes: bench.py at N = 1, 2, 4, 8 x exchange (RCCL gather | copy-engine IPC) x distribution (balanced strips | scanlines), with the
# display frame of every job compared with the N = 1 frame, and a table of what the display rank waits for.
# usage: bash tools/first_multi_gpu_run.sh [steps]      (from the repository root; writes gpurun_out/multi/)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${1:-50}
OUT=$R/gpurun_out/multi; mkdir -p "$OUT"
NDEV=$(python - <<'PY'
import ctypes
n = ctypes.c_int(0)
try:
    hip = ctypes.CDLL("libamdhip64.so")
except OSError:
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
print(n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0)
PY
)
echo "devices: $NDEV"
if [ "$NDEV" -lt 2 ]; then echo "fewer than two devices: nothing here can run (TRHIP_TEST_MULTI_DEVICE_REHEARSAL=1 python -m pytest tests/test_multi_device.py -m gpu rehearses the scripts on one)"; exit 2; fi

echo "== 1. the exchanges between real devices (tests/test_multi_device.py)"
python -m pytest tests/test_multi_device.py -m gpu -q -x -rs 2>&1 | tee "$OUT/test_multi_device.txt" | tail -15
grep -q " failed\|error" "$OUT/test_multi_device.txt" && echo "!! the transports are not correct on this node: the rates below are not to be quoted"

echo "== 2. N = 1 (the reference frame and rate)"
python bench.py --steps "$STEPS" --no-cpu-baseline --no-pmc --save-display "$OUT/display_n1.npy" > "$OUT/n1.json" 2> "$OUT/n1.err" || tail -5 "$OUT/n1.err"

echo "== 3. N x exchange x distribution"
PORT=29611
for n in 2 4 8; do
  [ "$n" -gt "$NDEV" ] && continue
  for ex in native ipc; do
    for st in auto scanline; do
      tag="n${n}_${ex}_${st}"; PORT=$((PORT + 1))
      timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$PORT" bench.py --gpus "$n" --steps "$STEPS" --warmup 5 \
        --exchange "$ex" --strategy "$st" --save-display "$OUT/display_$tag.npy" > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { echo "$tag FAILED"; tail -5 "$OUT/$tag.err"; }
    done
  done
done

python - "$OUT" <<'PY'
import glob, json, os, sys
import numpy as np
out = sys.argv[1]
def line(p):
    l = [x for x in open(p) if x.startswith("{")]
    return json.loads(l[-1]) if l else None
one = line(os.path.join(out, "n1.json"))
ref = np.load(os.path.join(out, "display_n1.npy")) if os.path.exists(os.path.join(out, "display_n1.npy")) else None
print("%-22s %10s %8s %10s %8s %14s %12s %s" % ("job", "Mray/s", "x N=1", "pipelined", "x N=1", "transport_wait", "expected x", "display == N=1"))
if one:
    print("%-22s %10.0f %8s %10.0f" % ("n1", one["value"], "1.00", one.get("value_pipelined") or 0))
for p in sorted(glob.glob(os.path.join(out, "n[248]_*.json"))):
    tag = os.path.basename(p)[:-5]
    r = line(p)
    if not r:
        print("%-22s no result" % tag); continue
    d = os.path.join(out, "display_%s.npy" % tag)
    same = (ref is not None and os.path.exists(d) and np.array_equal(np.load(d), ref))
    ph = (r.get("rank_phases") or {}).get("display_rank") or {}
    exp = r.get("scaling_expected_vs_one_gpu")
    print("%-22s %10.0f %8.2f %10.0f %8.2f %14s %12s %s" % (tag, r["value"], r["value"] / one["value"] if one else 0, r.get("value_pipelined") or 0,
          (r.get("value_pipelined") or 0) / one["value_pipelined"] if one and one.get("value_pipelined") else 0, ph.get("transport_wait_ms"), json.dumps(exp) if exp else "-", "bit-equal" if same else "DIFFERS"))
PY
