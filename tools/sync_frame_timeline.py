"""Where the time of a one-frame-at-a-time frame goes on the GPU, from a rocprofv3 kernel trace of a bench run
(`rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --no-pmc --no-cpu-baseline --sustained-frames 0`).
Busy stretches (at least one kernel running) separated by idle gaps longer than 10 us are taken as frames when they last 1-20 ms:
per frame the stretch, the idle gap before it (host synchronisation + enqueue of the next frame), the kernel time inside it and the
average number of kernels running.  usage: python tools/sync_frame_timeline.py DIR/t_kernel_trace.csv"""
import csv
import re
import sys
import numpy as np

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"]).group(1), r.get("Queue_Id", "0"))
        for r in csv.DictReader(open(sys.argv[1])) if re.search(r"k_[a-z_0-9]+", r["Kernel_Name"])]
rows.sort()
all_rows = rows
rows = [r[:3] for r in rows]
stretches = []      # [start, end, kernel ns, {kernel: ns}]
for s, e, k in rows:
    if stretches and s <= stretches[-1][1] + 10_000:
        st = stretches[-1]
        st[1] = max(st[1], e); st[2] += e - s; st[3][k] = st[3].get(k, 0) + e - s; st[4] += 1
    else:
        stretches.append([s, e, e - s, {k: e - s}, 1])
frames = [(i, st) for i, st in enumerate(stretches) if 1e6 <= st[1] - st[0] <= 20e6 and "k_raygen" in st[3] and "k_resolve" in st[3]]
# the one-frame-at-a-time region: consecutive frame stretches of similar length (the pipelined regions are single long stretches)
lens = np.array([st[1] - st[0] for _, st in frames]) / 1e6
med = float(np.median(lens))
sel = [(i, st) for (i, st), l in zip(frames, lens) if abs(l - med) < 0.25 * med]
gaps = [(stretches[i][0] - stretches[i - 1][1]) / 1e3 for i, _ in sel[1:] if i > 0]      # idle time before the frame (after whatever ran before it)
span = np.array([st[1] - st[0] for _, st in sel]) / 1e6
ksum = np.array([st[2] for _, st in sel]) / 1e6
print(f"frames found: {len(sel)} (median stretch {med:.3f} ms)")
print(f"GPU busy stretch per frame: mean {span.mean():.4f} ms, p50 {np.median(span):.4f}; idle gap between frames: mean {np.mean(gaps):.1f} us, p50 {np.median(gaps):.1f} us")
print(f"=> frame period {span.mean() + np.mean(gaps) / 1e3:.4f} ms; kernel time inside a frame {ksum.mean():.4f} ms = {ksum.mean() / span.mean():.2f} kernels running on average; launches per frame {np.mean([st[4] for _, st in sel]):.1f}")
per = {}
for _, st in sel:
    for k, v in st[3].items():
        per[k] = per.get(k, 0) + v
print("kernel ms per frame:", {k: round(v / len(sel) / 1e6, 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1])})

# per lane (= hardware queue): when its first kernel starts and its last one ends, as fractions of the frame's busy stretch, and how
# many kernels run in each tenth of the frame
lanes, tenths = {}, np.zeros(10)
for _, st in sel:
    f0, f1 = st[0], st[1]
    seen = {}
    for s0, e0, k, q in all_rows:
        if s0 < f0 or e0 > f1:
            continue
        a = seen.setdefault(q, [s0, e0, 0, 0])
        a[0] = min(a[0], s0); a[1] = max(a[1], e0); a[2] += e0 - s0; a[3] += 1
        for t in range(10):
            lo, hi = f0 + (f1 - f0) * t / 10, f0 + (f1 - f0) * (t + 1) / 10
            tenths[t] += max(0.0, min(e0, hi) - max(s0, lo)) / (hi - lo)
    for rank, (q, a) in enumerate(sorted(seen.items(), key=lambda kv: kv[1][0])):      # lanes in the order they start
        lanes.setdefault(rank, []).append(((a[0] - f0) / (f1 - f0), (a[1] - f0) / (f1 - f0), a[2] / (a[1] - a[0]), a[3]))
print("lanes in the order they start: first kernel starts at / last kernel ends at (fraction of the frame), busy fraction in between, launches")
for rank, v in sorted(lanes.items()):
    v = np.array(v)
    print(f"  lane {rank}: {v[:, 0].mean():.3f} .. {v[:, 1].mean():.3f}   busy {v[:, 2].mean():.2f}   launches {v[:, 3].mean():.1f}")
print("kernels running, by tenth of the frame:", " ".join(f"{x / len(sel):.2f}" for x in tenths))
