#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
W, H = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"
scene = scenes.WORKLOADS[name](W, H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene); print(ss.accel)
opt = R.options_for_scene(scene, max_bounces=4)
pt = R.PathTracerStage(ctx, ss, opt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
color = ctx.alloc(W * H * 16).zero()
pt.set_profiling(len(sys.argv) > 3, True)
for f in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    pt.reset_accumulated_samples(); pt.reset_counters()
    t = time.perf_counter(); pt.run(color); ctx.sync(); dt = time.perf_counter() - t
    tm = pt.timings(); c = pt.counters()
    img = color.download((H, W, 4))
    print(f"frame {f}: {dt*1e3:.2f} ms closest {tm['trace_closest_ms']:.2f} shadow {tm['trace_shadow_ms']:.2f} shade {tm['shade_ms']:.2f} rays {c['closest_rays']+c['shadow_rays']} nonfinite px {(~np.isfinite(img)).any(-1).sum()} mean {np.nanmean(img[...,:3]):.4f} max {np.nanmax(img[...,:3]):.1f}")
