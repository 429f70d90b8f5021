"""Bisecting experiments/r5_two_rays_per_lane.patch (round 6): where do its frames part from the one-ray kernel's?
usage (GPU box, libraries built from commit ddf40d6 + the patch):
  TRHIP_LIB=<lib> python tools/debug/rays2_bisect.py dump out.npz        frames of test.glb 96x96 under option sets that isolate parts of the path
  python tools/debug/rays2_bisect.py compare a.npz b.npz                 per option set: pixels that differ, by what the primary ray hit, and the sign
The option sets: bounces 1 (primary hits only), bounces 2 (one more closest-hit trace), bounces 2 without next-event estimation (no shadow rays:
only the two closest-hit traces decide the pixel), bounces 2 with min_ray_dist 0, bounces 2 with lights hidden from rays."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
SETS = {"bounces1": dict(max_bounces=1), "bounces2": dict(max_bounces=2),
        "bounces2_no_nee": dict(max_bounces=2, nee_point=0.0, nee_directional=0.0, nee_envmap=0.0, nee_triangles=0.0),
        "bounces2_tmin0": dict(max_bounces=2, min_ray_dist=0.0), "bounces2_hidden_lights": dict(max_bounces=2, hide_lights=1),
        "bounces3_no_nee": dict(max_bounces=3, nee_point=0.0, nee_directional=0.0, nee_envmap=0.0, nee_triangles=0.0)}
if sys.argv[1] == "compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    inst = a["bounces1_instance_id"][0, ..., 0]
    for name in SETS:
        x, y = a[name + "_color"][0, ..., :3].astype(np.float64), b[name + "_color"][0, ..., :3].astype(np.float64)
        bad = (a[name + "_color"].view(np.uint32) != b[name + "_color"].view(np.uint32)).any(-1)[0]
        print(f"{name}: {int(bad.sum())} of {bad.size} pixels differ; mean {x.mean():.6f} vs {y.mean():.6f}")
        if bad.any():
            lum = (y - x).sum(-1)[bad]
            ids, cnt = np.unique(inst[bad], return_counts=True)
            print(f"    second library brighter in {int((lum > 0).sum())}, darker in {int((lum < 0).sum())}; |difference| median {np.median(np.abs(lum)):.4g}, max {np.abs(lum).max():.4g}")
            print("    by instance the primary ray hit (-1 = light or miss):", dict(zip(ids.tolist(), cnt.tolist())))
            ys, xs = np.nonzero(bad)
            for yy, xx in list(zip(ys, xs))[:5]:
                print("    pixel", yy, xx, "instance", int(inst[yy, xx]), x[yy, xx], y[yy, xx])
    sys.exit(0)
from tauray_amd import renderer as R
from tauray_amd.gltf import load_glb
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
W = H = 96
scene = load_glb(os.path.join(root, "tests", "golden", "test.glb"), W, H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene)
out = {}
for name, kw in SETS.items():
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, **kw), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    names = ["color", "instance_id"]
    bufs = {n: ctx.alloc(W * H * R.PathTracerStage.TARGETS[n][0] * 4).zero() for n in names}
    pt.run_targets(bufs)
    for n in names:
        ch, dt = R.PathTracerStage.TARGETS[n]
        out[f"{name}_{n}"] = np.frombuffer(bufs[n].download((1, H, W, ch)).tobytes(), dtype=dt).reshape(1, H, W, ch)
    pt.close()
np.savez(sys.argv[2], **out)
print("wrote", sys.argv[2], "point lights:", len(scene.point_lights), "radii", [float(p["radius"]) if "radius" in p.dtype.names else None for p in scene.point_lights][:4])
