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
def work():
    """node visits and triangle tests per ray of one frame (the counting kernels)"""
    pt.set_profiling(True, False); pt.reset_counters()
    pt.reset_accumulated_samples(); pt.run(color); ctx.sync()
    c = pt.counters(); pt.set_profiling(False, False)
    rays = c["closest_rays"] + c["shadow_rays"]
    return f"{c['node_visits'] / rays:.2f} node visits, {c['tri_tests'] / rays:.2f} triangle tests per ray"
print("frame on the static tree: %.3f ms; %s" % (frame_ms(), work()))
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
    print(f"{mode:>15}: {host:7.2f} ms per update on the host clock ({dev:.2f} ms of it in the library's own build timer); frame afterwards {frame_ms():.3f} ms; {work()}")
    ss.update_instances(inst, refit=False)      # back to the first pose, rebuilt
# the first pose again, rebuilt both ways: is a rebuilt tree as good as the first one?
for static in (True, False, True):
    ss.fast_trace_rebuilds = static
    acc = ss.update_instances(inst, refit=False)
    print(f"first pose rebuilt ({'static' if static else 'fast'}): build {acc['build_ms']:.2f} ms, {acc['node_count']} nodes, frame {frame_ms():.3f} ms; {work()}")
ss2 = R.SceneStage(ctx, sc)
pt2 = R.PathTracerStage(ctx, ss2, opt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
pt_old, pt = pt, pt2
print(f"a second scene stage built from scratch: build {ss2.accel['build_ms']:.2f} ms, {ss2.accel['node_count']} nodes, frame {frame_ms():.3f} ms")
