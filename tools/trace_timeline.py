"""Where the clocks of a closest-hit wave go, phase by phase (csrc/trace_timeline.h).
usage (on a GPU box):  TRHIP_LIB=tauray_amd/libtrhip_timeline.so python tools/trace_timeline.py [workload] [frames] > profiles/r5/trace_phase_timeline.txt
The library is the variant built by `make -C tauray_amd/csrc variant NAME=timeline EXTRA=-DTR_TIMELINE=1`.  Frames run with detailed
timing, i.e. one lane on one stream, every kernel alone on the chip: the instance timed is k_trace_closest<false, true, ..>, the
roofline kernel of bench.py."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tauray_amd import renderer as R, scenes, _lib   # noqa: E402
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
lib = _lib.lib()
if not hasattr(lib, "trhip_debug_timeline"):
    sys.exit("this library has no phase timeline: build the variant with -DTR_TIMELINE=1 and select it with TRHIP_LIB")
scene = scenes.WORKLOADS[wl](W, H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
pt.set_profiling(False, True)
color = ctx.alloc(W * H * 16).zero()
for _ in range(3):
    pt.run(color)
ctx.sync() if hasattr(ctx, "sync") else None
ROWS, NB = 28, 8
buf = (C.c_uint64 * (ROWS * NB))()
lib.trhip_debug_timeline(None, 1)
pt.reset_counters()
for _ in range(frames):
    pt.reset_accumulated_samples()
    pt.run(color)
assert lib.trhip_debug_timeline(buf, 0) == 0
t = pt.timings()
tl = [[int(buf[r * NB + b]) for b in range(NB)] for r in range(ROWS)]
misc = tl[20]
chunks, clocks_in, deal, fetch, wall, chunk_clocks = misc[0], misc[1], misc[2], misc[3], misc[4], misc[5]
ghz = clocks_in / (wall * 10.0) if wall else 0.0     # s_memrealtime ticks at 100 MHz
print(f"# phase timeline of k_trace_closest, {wl} {W}x{H}, {frames} frames, 4 bounces; timeline build of the library")
print(f"# closest-hit kernel: {t['trace_closest_ms'] / max(1, t['trace_closest_launches']):.4f} ms per launch under the instrument "
      f"({t['trace_closest_launches']} launches; the production library's figure is in the bench line of the same session)")
print(f"# s_memtime ticks per 100 MHz tick: {clocks_in / wall if wall else 0:.2f}  -> stamp clock {ghz * 1000:.0f} MHz")
print(f"# chunks of 64 rays {chunks}; clocks of a wave inside the traversal per chunk {clocks_in / chunks:.0f}; "
      f"per chunk with the ray fetch and the hit store {chunk_clocks / chunks:.0f}; waiting for the rays {fetch / chunks:.0f}; re-deal to quads {deal / chunks:.0f}")


def table(name, row, units, bucket_names):
    print(f"\n## {name}")
    print(f"{'bucket':>10} {'phases':>10} {'/chunk':>7} {units + '/ph':>8} {'issue':>7} {'wait':>7} {'compute':>8} {'total':>7} {'share':>6}")
    tot = [0] * 5
    for b in range(NB):
        n = tl[row][b]
        if not n:
            continue
        u, i, w, c = tl[row + 1][b], tl[row + 2][b], tl[row + 3][b], tl[row + 4][b]
        for k, v in enumerate((n, u, i, w, c)):
            tot[k] += v
        print(f"{bucket_names[b]:>10} {n:>10} {n / chunks:>7.2f} {u / n:>8.1f} {i / n:>7.0f} {w / n:>7.0f} {c / n:>8.0f} {(i + w + c) / n:>7.0f} {(i + w + c) / clocks_in:>6.1%}")
    n, u, i, w, c = tot
    if n:
        print(f"{'all':>10} {n:>10} {n / chunks:>7.2f} {u / n:>8.1f} {i / n:>7.0f} {w / n:>7.0f} {c / n:>8.0f} {(i + w + c) / n:>7.0f} {(i + w + c) / clocks_in:>6.1%}")
    return i + w + c, i, w, c


lane_b = [f"{8 * b + 1}-{8 * b + 8}" for b in range(NB)]
quad_b = [f"{2 * b + 1}-{2 * b + 2}" for b in range(NB)]
print("\n(clocks per phase of one wave: issue = phase start to the last vector load issued; wait = s_waitcnt vmcnt(0); compute = slab tests, sort,\n"
      " stack pushes and pop / triangle test, candidate, alpha; share = of a wave's clocks inside the traversal)")
a = table("per-lane loop, node phases by lanes taking part", 0, "lanes", lane_b)
b_ = table("per-lane loop, triangle phases by lanes taking part", 5, "lanes", lane_b)
c_ = table("quad tail, node phases by quads taking part", 10, "quads", quad_b)
d_ = table("quad tail, triangle phases by lanes holding a triangle", 15, "lanes", lane_b)
booked = a[0] + b_[0] + c_[0] + d_[0] + deal
print(f"\n## a wave's clocks inside the traversal: {clocks_in / chunks:.0f} per chunk")
for name, v in (("node phases, per-lane", a), ("triangle phases, per-lane", b_), ("node phases, quads", c_), ("triangle phases, quads", d_)):
    print(f"{name:>28}: {v[0] / clocks_in:6.1%}   (issue {v[1] / clocks_in:5.1%}, wait {v[2] / clocks_in:5.1%}, compute {v[3] / clocks_in:5.1%})")
print(f"{'re-deal to quads':>28}: {deal / clocks_in:6.1%}")
print(f"{'votes, ballots, loop, setup':>28}: {(clocks_in - booked) / clocks_in:6.1%}")
wt = a[2] + b_[2] + c_[2] + d_[2]
print(f"{'all waiting for loads':>28}: {wt / clocks_in:6.1%};  all issue {(a[1] + b_[1] + c_[1] + d_[1]) / clocks_in:6.1%};  all compute {(a[3] + b_[3] + c_[3] + d_[3]) / clocks_in:6.1%}")
bins = ["<250", "<500", "<1000", "<1500", "<2000", "<3000", "<4000", ">=4000"]
for name, row in (("per-lane node phases", 21), ("quad node phases", 22), ("per-lane triangle phases", 23)):
    tot = sum(tl[row]) or 1
    print(f"\n## wait clocks of {name} (share of phases): " + "  ".join(f"{bins[k]} {tl[row][k] / tot:.1%}" for k in range(NB)))

print("\n## compute of the per-lane phases, split (clocks per phase)")
print(f"{'bucket':>10} {'node: slab+sort':>16} {'pushes+pop':>11} | {'tri: intersect':>15} {'candidate/any-hit':>18} {'pop':>6} {'phases with an alpha test':>27}")
for b in range(NB):
    nn, nt = tl[0][b], tl[5][b]
    if not (nn or nt):
        continue
    ns = tl[24][b] / nn if nn else 0
    nc = tl[4][b] / nn if nn else 0
    ti, ta = (tl[25][b] / nt, tl[26][b] / nt) if nt else (0, 0)
    tc = tl[9][b] / nt if nt else 0
    print(f"{lane_b[b]:>10} {ns:>16.0f} {nc - ns:>11.0f} | {ti:>15.0f} {ta:>18.0f} {tc - ti - ta:>6.0f} {(tl[27][b] / nt if nt else 0):>27.1%}")
