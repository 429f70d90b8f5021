"""First-hit targets (instance id, position, normal) and the colour of a small test.glb frame for the library TRHIP_LIB selects, as .npz: two
builds that promise the same hits are compared field by field.  usage: TRHIP_LIB=... python tools/debug/first_hit_dump.py out.npz [bounces];
python tools/debug/first_hit_dump.py --compare a.npz b.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        bad = (a[k].view(np.uint32) != b[k].view(np.uint32)).any(-1)
        print(f"{k}: {int(bad.sum())} of {bad.size} pixels differ")
        ys, xs = np.nonzero(bad[0])
        for y, x in list(zip(ys, xs))[:4]:
            print("   ", y, x, a[k][0, y, x], b[k][0, y, x])
    sys.exit(0)
from tauray_amd import renderer as R
from tauray_amd.gltf import load_glb
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
W = H = 96
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
scene = load_glb(os.path.join(root, "tests", "golden", "test.glb"), W, H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=int(sys.argv[2]) if len(sys.argv) > 2 else 2), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
names = ["color", "instance_id", "pos", "normal", "diffuse", "reflection"]
bufs = {n: ctx.alloc(W * H * R.PathTracerStage.TARGETS[n][0] * 4).zero() for n in names}
pt.run_targets(bufs)
out = {}
for n in names:
    ch, dt = R.PathTracerStage.TARGETS[n]
    out[n] = np.frombuffer(bufs[n].download((1, H, W, ch)).tobytes(), dtype=dt).reshape(1, H, W, ch)
np.savez(sys.argv[1], **out)
