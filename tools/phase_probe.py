"""How long the tails of the closest-hit waves are: node phases of the per-lane traversal loop by number of active rays
(counting build, TRHIP_DEBUG prints the statistics).  usage: python tools/phase_probe.py [workload] [frames]"""
import os, sys
os.environ["TRHIP_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
wl = sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = 1920, 1080
scene = scenes.WORKLOADS[wl](W, H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
pt.set_profiling(True, False)
color = ctx.alloc(W * H * 16).zero()
for _ in range(frames):
    pt.reset_accumulated_samples()
    pt.run(color)
c = pt.counters()
print(wl, {k: v // frames for k, v in c.items()})
