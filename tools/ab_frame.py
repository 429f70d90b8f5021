"""One 1080p frame (4 bounces, the reference's command-line options) of a bench scene, saved as .npy, for the library TRHIP_LIB selects: two builds
that promise the same results are compared bit by bit.  usage: TRHIP_LIB=... python tools/ab_frame.py out.npy [sponza_teapots|sponza_class]; then
python tools/ab_frame.py --compare a.npy b.npy"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    diff = (a.view(np.uint32) != b.view(np.uint32)).any(-1)
    print(f"{sys.argv[2]} vs {sys.argv[3]}: {int(diff.sum())} of {diff.size} pixels differ; mean {float(a[..., :3].mean()):.6f}")
    sys.exit(1 if diff.any() else 0)
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
W, H = 1920, 1080
scene = getattr(scenes, sys.argv[2] if len(sys.argv) > 2 else "sponza_teapots")(width=W, height=H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
buf = ctx.alloc(W * H * 16).zero()
pt.run(buf)
assert pt.counters()["stack_overflows"] == 0
np.save(sys.argv[1], buf.download((H, W, 4)))
