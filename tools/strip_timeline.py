"""The kernels of one rank's share of a frame, one frame at a time, as a timeline: which launch runs when on which hardware queue.
  render:  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/strip_timeline.py render [workload] [world] [frames]
  report:  python tools/strip_timeline.py report DIR/.../t_kernel_trace.csv > profiles/r5/strip_timeline_1_8.txt
The render leg draws the last rank's share of a job of `world` ranks (shuffled strips) `frames` times with a host sync after every
frame; the report leg takes the median frame of the trace and prints its launches per queue with start offsets and durations."""
import os
import sys

if sys.argv[1] == "render":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tauray_amd import renderer as R, scenes
    from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS
    wname = sys.argv[2] if len(sys.argv) > 2 else "sponza_teapots"
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    W, H = 1920, 1080
    ctx = R.Context(0)
    sc = scenes.WORKLOADS[wname](W, H)
    opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
    rr = R.RtRenderer(ctx, sc, opt, (W, H), strategy=DISTRIBUTION_SHUFFLED_STRIPS, rank=world - 1, world_size=world, use_torch=False)
    import time
    for i in range(n + 10):
        if i == 10:
            t0 = time.perf_counter()
        rr.reset_accumulation(); rr.render_partial(); rr.sync()
    print(f"{wname} 1/{world}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per frame (host wall time, with or without the tracer around it)", file=sys.stderr)
    rr.close()
    sys.exit(0)

import csv
import re
import numpy as np
def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else re.sub(r"[^A-Za-z_0-9].*", "", name)[:18]


rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0"),
         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
        for r in csv.DictReader(open(sys.argv[2]))]
rows.sort()
stretches = []
for s, e, k, q, g, b in rows:
    if stretches and s <= stretches[-1][1] + 30_000:      # a frame starts host-bound: up to ~25 us between its first kernel and the next
        stretches[-1][1] = max(stretches[-1][1], e); stretches[-1][2].append((s, e, k, q, g, b))
    else:
        stretches.append([s, e, [(s, e, k, q, g, b)]])
frames = [st for st in stretches if any(x[2] == "k_raygen" for x in st[2]) and any(x[2] == "k_resolve" for x in st[2]) and st[1] - st[0] < 20e6]
frames = frames[len(frames) // 3:]
lens = np.array([st[1] - st[0] for st in frames])
med = frames[int(np.argsort(lens)[len(lens) // 2])]
gaps = [b[0] - a[1] for a, b in zip(frames[:-1], frames[1:])]
print(f"# frames in the trace {len(frames)}; busy stretch p50 {np.median(lens) / 1e3:.1f} us, mean {lens.mean() / 1e3:.1f}; idle before the next frame p50 {np.median(gaps) / 1e3:.1f} us")
print(f"# the median frame: {(med[1] - med[0]) / 1e3:.1f} us, {len(med[2])} launches")
queues = sorted({x[3] for x in med[2]}, key=lambda q: min(x[0] for x in med[2] if x[3] == q))
for q in queues:
    print(f"\n## queue {q}")
    prev = None
    print(f"{'kernel':>18} {'start us':>9} {'dur us':>8} {'gap us':>7} {'blocks':>7}")
    for s, e, k, _, g, b in [x for x in med[2] if x[3] == q]:
        print(f"{k:>18} {(s - med[0]) / 1e3:>9.1f} {(e - s) / 1e3:>8.1f} {((s - prev) / 1e3 if prev else 0):>7.1f} {g // max(1, b):>7}")
        prev = e
# over all frames: per kernel name mean duration and mean gap to the previous launch of the same queue
agg, gapagg = {}, []
for st in frames:
    last = {}
    for s, e, k, q, g, b in st[2]:
        agg.setdefault(k, []).append(e - s)
        if q in last:
            gapagg.append(s - last[q])
        last[q] = e
print("\n## all frames: mean duration per launch (us), launches per frame")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:>18} {np.mean(v) / 1e3:>8.1f} {len(v) / len(frames):>6.1f}   per frame {sum(v) / len(frames) / 1e3:>8.1f}")
print(f"gap between consecutive launches of a queue: mean {np.mean(gapagg) / 1e3:.1f} us, p50 {np.median(gapagg) / 1e3:.1f}, p90 {np.percentile(gapagg, 90) / 1e3:.1f}")
