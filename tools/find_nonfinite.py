"""Finds the first frames of a long accumulation that produce a non-finite pixel, and asks the oracle about the same pixel.
usage: python tools/find_nonfinite.py [workload] [frames] [width height]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE

wl = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
scene = scenes.WORKLOADS[wl](W, H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene)
d = DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4), d)
acc = ctx.alloc(W * H * 16).zero()
STEP = 64
bad_frames = []
for f0 in range(0, N, STEP):
    pt.reset_accumulated_samples()
    pt.set_frame_counter(f0)
    for _ in range(STEP):
        pt.run(acc)
    img = acc.download((H, W, 4))
    if not np.isfinite(img).all() or (img[..., :3] < 0).any() or (img[..., 3] != 1).any():
        for f in range(f0, f0 + STEP):
            pt.reset_accumulated_samples()
            pt.set_frame_counter(f)
            pt.run(acc)
            one = acc.download((H, W, 4))
            bad = ~np.isfinite(one).all(-1) | (one[..., :3] < 0).any(-1) | (one[..., 3] != 1)
            if bad.any():
                ys, xs = np.nonzero(bad)
                print(f"frame {f}: {bad.sum()} bad pixels, first at x={xs[0]} y={ys[0]} value={one[ys[0], xs[0]]}", flush=True)
                bad_frames.append((f, int(xs[0]), int(ys[0])))
    if len(bad_frames) >= 3:
        break
print("bad frames:", bad_frames)
if bad_frames:
    from oracle import binding as B
    osc = B.OracleScene(scene)
    oopt = B.options_for_scene(scene, max_bounces=4)
    for f, x, y in bad_frames[:2]:
        ref = osc.render_pt(oopt, W, H, frame_counter=f)[0]
        print(f"oracle frame {f} pixel ({x},{y}) = {ref[y, x]}; non-finite pixels in the oracle frame: {int((~np.isfinite(ref).all(-1)).sum())}")
