"""One in-process multi-rank job (tests/test_gpu_parity.py _in_process_job) against the single-rank frames, for a list of
configurations W H world strategy F B: which frames differ and by how many pixels.  usage (through gpurun):
python tools/debug/in_process_job_debug.py "62 18 1 2 4 4" "62 18 1 2 1 1" ..."""
import os, sys
import numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_parity as T
from tauray_amd import renderer as R
from tauray_amd.gltf import load_glb
ctx = R.Context(0)
for spec in sys.argv[1:]:
    W, H, world, strategy, F, B = [int(x) for x in spec.split()]
    frames = B * 3
    scene = load_glb(os.path.join(root, "tests", "golden", "test.glb"), W, H)
    opt = R.options_for_scene(scene, max_bounces=2)
    ref = T._single_rank_frames(R, ctx, scene, opt, (W, H), frames)
    got = T._in_process_job(R, scene, opt, (W, H), world, strategy, F, frames, B=B)
    diff = [(f, int((got[f] != ref[f]).any(-1).sum())) for f in range(frames)]
    print(spec, "->", diff, "| max abs diff", float(np.abs(got - ref).max()))
    if any(n for _, n in diff):
        f = [f for f, n in diff if n][0]
        ys, xs = np.nonzero((got[f] != ref[f]).any(-1))
        print("   first differing frame", f, "rows", sorted(set(ys.tolist()))[:12], "cols", sorted(set(xs.tolist()))[:12], "... got", got[f][ys[0], xs[0]], "ref", ref[f][ys[0], xs[0]])
