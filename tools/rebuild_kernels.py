"""Ten fast rebuilds of the bench scene's acceleration structure (for `rocprofv3 --kernel-trace --stats`: which kernels a rebuild spends its
time in).  usage (through gpurun): rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o t -- python tools/rebuild_kernels.py"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from tauray_amd import renderer as R, scenes
ctx = R.Context(0)
scene = scenes.sponza_teapots(width=64, height=64)
ss = R.SceneStage(ctx, scene); ctx.sync()
ss.fast_trace_rebuilds = False
for _ in range(10):
    ss.update_instances(scene.instances, refit=False)
ctx.sync()
