"""Scale check: the atrium generator at 4, 16 and 48 million triangles - upload, on-device build, one 1080p frame of 4 bounces with counters,
a second build as a refit; prints triangles, build / refit / frame times, node visits per ray, device memory in use and the overflow
flag.  usage (through gpurun): python tools/debug/big_scene_probe.py [millions ...]"""
import os, sys, time
import numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, root)
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
import ctypes as C
hip = C.CDLL("libamdhip64.so")
def mem():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return (t.value - f.value) / 2**30
ctx = R.Context(0)
W, H = 1920, 1080
for millions in [float(x) for x in (sys.argv[1:] or ["4", "16", "48"])]:
    t0 = time.time()
    scene = scenes.sponza_class(seed=1, target_tris=int(millions * 1e6), teapots=50, width=W, height=H)
    t_gen = time.time() - t0
    t0 = time.time(); ss = R.SceneStage(ctx, scene); ctx.sync(); t_up = time.time() - t0
    opt = R.options_for_scene(scene, max_bounces=4)
    pt = R.PathTracerStage(ctx, ss, opt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    pt.set_profiling(True, False)
    buf = ctx.alloc(W * H * 16).zero()
    pt.run(buf); ctx.sync()
    t0 = time.time()
    for _ in range(5):
        pt.reset_accumulated_samples(); pt.run(buf)
    ctx.sync(); t_frame = (time.time() - t0) / 5
    c = pt.counters()
    img = buf.download((H, W, 4))
    prod = R.PathTracerStage(ctx, ss, opt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))      # the production kernels
    for _ in range(3):
        prod.reset_accumulated_samples(); prod.run(buf)
    ctx.sync(); t0 = time.time()
    for _ in range(20):
        prod.reset_accumulated_samples(); prod.run(buf)
    ctx.sync(); t_prod = (time.time() - t0) / 20
    prod.close()
    t0 = time.time(); ss.update_instances(scene.instances, refit=True); ctx.sync(); t_refit = time.time() - t0
    print(f"{scene.spans['triangle_count'].sum() / 1e6:.2f} M triangles: generated in {t_gen:.1f} s, upload + build {t_up:.2f} s ({ss.accel}), refit {t_refit * 1e3:.1f} ms, "
          f"frame {t_prod * 1e3:.2f} ms (counting instance {t_frame * 1e3:.2f} ms), {(c['closest_rays'] + c['shadow_rays']) / 5 / 1e6:.2f} M rays, {c['node_visits'] / max(c['closest_rays'] + c['shadow_rays'], 1):.2f} visits/ray, "
          f"overflow {c['stack_overflows']}, finite {bool(np.isfinite(img).all())}, mean {float(img[..., :3].mean()):.4f}, device memory in use {mem():.1f} GiB", flush=True)
    pt.close(); del ss
