"""On-device acceleration-structure times by scene size: the static build (PLOC + reinsertion + cost-based collapse; what a scene gets once), the
fast rebuild of a dynamic scene (src/acceleration_structure.cc:129-131's ePreferFastBuild), a refit; wall time around the call with a
device sync, best of five, and the frame the resulting tree gives.  usage (through gpurun): python tools/debug/build_time_probe.py [millions ...]"""
import os, sys, time
import numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, root)
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
ctx = R.Context(0)
W, H = 1920, 1080
def frame_ms(ss, scene):
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    buf = ctx.alloc(W * H * 16).zero()
    for _ in range(3):
        pt.reset_accumulated_samples(); pt.run(buf)
    ctx.sync(); t0 = time.time()
    for _ in range(20):
        pt.reset_accumulated_samples(); pt.run(buf)
    ctx.sync(); t = (time.time() - t0) / 20
    pt.close()
    return t * 1e3
for millions in [float(x) for x in (sys.argv[1:] or ["0.26", "1", "4"])]:
    scene = scenes.sponza_class(seed=1, target_tris=int(millions * 1e6), teapots=50 if millions >= 1 else 0, width=W, height=H)
    ss = R.SceneStage(ctx, scene); ctx.sync()
    static_frame = frame_ms(ss, scene)
    def best(fn, n=5):
        ts = []
        for _ in range(n):
            ctx.sync(); t0 = time.time(); fn(); ctx.sync(); ts.append(time.time() - t0)
        return min(ts) * 1e3
    t_static = best(lambda: ss.set_scene(scene), 3)          # upload + static build
    ss.fast_trace_rebuilds = False
    t_fast = best(lambda: ss.update_instances(scene.instances, refit=False))
    fast_frame = frame_ms(ss, scene)
    t_refit = best(lambda: ss.update_instances(scene.instances, refit=True))
    print(f"{scene.spans['triangle_count'].sum() / 1e6:.2f} M triangles: upload + static build {t_static:.1f} ms (frame {static_frame:.2f} ms); "
          f"fast rebuild {t_fast:.1f} ms (frame {fast_frame:.2f} ms); refit {t_refit:.2f} ms", flush=True)
    del ss
