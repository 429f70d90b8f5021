#!/usr/bin/env python3
"""Ad-hoc GPU bring-up check (run through gpurun): HIP path vs CPU oracle on test.glb."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tauray_amd.gltf import load_glb
from tauray_amd import renderer as R
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
from oracle import binding as B

W = H = 512
scene = load_glb(os.path.join(ROOT, "tests/golden/test.glb"), W, H)
ctx = R.Context(0)
t = time.time(); ss = R.SceneStage(ctx, scene); print("upload+build", time.time() - t, ss.accel)
osc = B.OracleScene(scene)
# tri lights
tl_g, tl_o = ss.tri_lights(), osc.tri_lights()
print("tri lights equal:", np.array_equal(tl_g.view(np.uint8), tl_o.view(np.uint8)), len(tl_g))
# features vs golden and oracle
names = {5: 'distance', 3: 'world-pos', 4: 'view-pos', 1: 'world-normal', 2: 'view-normal', 0: 'albedo'}
dist = DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True)
for f, n in names.items():
    fs = R.FeatureStage(ctx, ss, f, dist)
    buf = ctx.alloc(W * H * 16).zero()
    fs.run(buf)
    img = buf.download((H, W, 4))
    g = np.load(os.path.join(ROOT, f'tests/golden/validate_{n}.npz'))['rgb'].astype(np.float32)
    o = osc.render_feature(f, W, H)
    d = np.abs(img[..., :3] - g)
    tol = np.abs(g) * 2**-10 + 2e-3
    print(n, 'vs golden: bad px', int((d > tol).any(-1).sum()), 'max', float(d.max()), '| vs oracle: max abs', float(np.abs(img - o).max()),
          'bit-equal px %.4f' % float((img == o).all(-1).mean()))
# path tracer 1spp
for mb in (1, 2, 4, 8):
    opt = R.options_for_scene(scene, max_bounces=mb)
    pt = R.PathTracerStage(ctx, ss, opt, dist)
    pt.set_profiling(True, False)
    color = ctx.alloc(W * H * 16).zero()
    pt.run(color); ctx.sync()
    t = time.time(); pt.reset_accumulated_samples(); pt.reset_sample_counter(); pt.reset_counters(); pt.run(color); ctx.sync(); dt = time.time() - t
    img = color.download((H, W, 4))
    ref = osc.render_pt(B.options_for_scene(scene, max_bounces=mb), W, H)[0]
    rel = np.abs(img[..., :3] - ref[..., :3]) / (np.abs(ref[..., :3]) + 1e-2)
    c = pt.counters()
    print(f"bounces {mb}: finite {np.isfinite(img).all()} mismatch px {(rel.max(-1) > 1e-2).mean():.4%} exact px {(img == ref).all(-1).mean():.4%} "
          f"mean hip {img[..., :3].mean():.6f} oracle {ref[..., :3].mean():.6f} | {dt*1e3:.2f} ms, rays {c['closest_rays'] + c['shadow_rays']}, "
          f"{(c['closest_rays'] + c['shadow_rays']) / dt / 1e6:.1f} Mray/s", c, pt.timings())
