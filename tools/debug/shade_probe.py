import sys, time
sys.path.insert(0, '.')
import numpy as np
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
ctx = R.Context(0)
W, H = 1920, 1080
for wname in ("sponza_teapots",):
    sc = scenes.WORKLOADS[wname](W, H)
    ss = R.SceneStage(ctx, sc)
    variants = {
        "default": {},
        "no-nee": dict(nee_point=0.0, nee_directional=0.0, nee_triangles=0.0, nee_envmap=0.0),
        "nee-dir-only": dict(nee_point=0.0, nee_triangles=0.0, nee_envmap=0.0),
        "tri-area": dict(tri_light_mode=0),
        "bounce-cosine": dict(bounce_mode=1),
        "bounce-hemisphere": dict(bounce_mode=0),
        "mis-off": dict(mis_mode=0),
    }
    for name, kw in variants.items():
        opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1, **kw)
        pt = R.PathTracerStage(ctx, ss, opt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
        color = ctx.alloc(W * H * 16).zero()
        pt.set_profiling(False, True)
        for _ in range(3): pt.run(color)
        ctx.sync(); pt.reset_counters()
        for _ in range(10): pt.reset_accumulated_samples(); pt.run(color)
        ctx.sync()
        t = pt.timings(); c = pt.counters()
        print(f"{wname:15s} {name:18s} shade {t['shade_ms']/10:.3f} closest {t['trace_closest_ms']/10:.3f} shadow {t['trace_shadow_ms']/10:.3f} rays {(c['closest_rays']+c['shadow_rays'])/10/1e6:.2f}M shadow {c['shadow_rays']/10/1e6:.2f}M")
        pt.close()
