"""Dynamic scenes: what an instance update costs per frame on the bench scene - refit (tree kept, boxes recomputed),
fast rebuild, static rebuild - and what the frame costs on the tree each leaves behind when the teapots have moved."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
W, H = 1920, 1080
ctx = R.Context(0)
sc = scenes.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"](W, H)
ss = R.SceneStage(ctx, sc)
print("first build (static):", round(ss.accel["build_ms"], 2), "ms;", ss.accel["node_count"], "nodes")
opt = R.options_for_scene(sc, max_bounces=4)
pt = R.PathTracerStage(ctx, ss, opt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
color = ctx.alloc(W * H * 16).zero()
def frame_ms(n=30):
    for _ in range(5):
        pt.reset_accumulated_samples(); pt.run(color)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        pt.reset_accumulated_samples(); pt.run(color); ctx.sync()
    return (time.perf_counter() - t0) / n * 1e3
print("frame on the static tree: %.3f ms" % frame_ms())
inst = sc.instances.copy()
n_inst = len(inst)
print("instances:", n_inst)

def moved(step):
    """every instance but the first (the building) drifts along x by 2 cm per step (column-major model matrix: translation in model[3][:3])"""
    out = inst.copy()
    out["model_prev"] = out["model"]
    out["model"][1:, 3, 0] += 0.02 * step
    return out

for mode in ("refit", "fast rebuild", "static rebuild"):
    ss.fast_trace_rebuilds = (mode == "static rebuild")
    times = []
    for step in range(1, 9):
        t0 = time.perf_counter()
        acc = ss.update_instances(moved(step), refit=(mode == "refit"))
        ctx.sync()
        times.append(((time.perf_counter() - t0) * 1e3, acc["build_ms"]))
    host = np.median([t[0] for t in times]); dev = np.median([t[1] for t in times])
    print(f"{mode:>15}: {host:7.2f} ms per update on the host clock ({dev:.2f} ms of it in the library's own build timer); frame afterwards {frame_ms():.3f} ms")
    ss.update_instances(inst, refit=False)      # back to the first pose, rebuilt
