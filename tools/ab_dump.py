"""Dumps closest-hit query results and a small frame for the library selected by TRHIP_LIB (A/B debugging of trace kernels).
usage: TRHIP_LIB=... python tools/ab_dump.py out.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tauray_amd import renderer as R
from tauray_amd.gltf import load_glb
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
W = H = 96
scene = load_glb(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "test.glb"), W, H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene)
rng = np.random.default_rng(5)
n = 200_000
lo, hi = np.array(ss.accel["bounds_min"], np.float32), np.array(ss.accel["bounds_max"], np.float32)
org = (lo + (hi - lo) * rng.uniform(0.05, 0.95, size=(n, 3))).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d, np.full((n, 1), np.inf, np.float32)], axis=1)
seeds = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
out = {}
for tag, s in (("seeded", seeds), ("fixed", None)):
    g = ss.trace_closest(rays, s)
    for k in ("instance_id", "primitive_id", "t", "bary_u", "bary_v"):
        out[f"{tag}_{k}"] = np.asarray(g[k])
for bounces in (1, 2, 4):
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=bounces), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    color = ctx.alloc(W * H * 16).zero()
    pt.run(color)
    out[f"frame{bounces}"] = color.download((H, W, 4))
    pt.close()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1])
