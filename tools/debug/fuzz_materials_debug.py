"""Reproduces draw K of tests/test_gpu_parity.py::test_random_materials under fuzz seed S and prints the pixels where HIP (both shading\narithmetic modes) and the oracle differ most.  usage (through gpurun): python tools/debug/fuzz_materials_debug.py S K"""
import copy, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import test_gpu_parity as T
from tauray_amd import renderer as R, scene as S
from oracle import binding as oracle
ctx = R.Context(0)
seed, target = int(sys.argv[1]), int(sys.argv[2])
base = T._zoo_scene()
rng = np.random.default_rng(seed)
corner = lambda: float(rng.choice([0.0, 1.0, rng.uniform(0, 1), rng.uniform(0, 1)]))
for k in range(target + 1):
    sc = copy.copy(base)
    sc.instances = base.instances.copy()
    mats = []
    for i in range(9):
        ior = float(rng.uniform(0.5, 3.0))
        if abs(ior - 1.0) < 0.03:
            ior = 1.3
        emis = tuple(rng.uniform(0, 4, 3)) if rng.uniform() < 0.25 else (0, 0, 0)
        a = dict(albedo=tuple(rng.uniform(0, 1, 3)) + (float(rng.choice([1.0, 1.0, rng.uniform(0.1, 0.9)])),), metallic=corner(), roughness=corner(), emission=emis,
                 transmittance=float(rng.choice([0.0, 0.0, 1.0, rng.uniform(0, 1)])), ior=ior, normal_factor=float(rng.uniform(0.5, 1.5)), double_sided=bool(rng.integers(0, 2)))
        mats.append(a)
        sc.instances["mat"][i] = S.make_material(**a)
    sc.finalize(True)
    kw = dict(max_bounces=int(rng.integers(2, 6)), sampler=int(rng.integers(0, 2)), tri_light_mode=int(rng.integers(0, 3)))
    if k < target:
        continue
    ss = R.SceneStage(ctx, sc)
    osc = oracle.OracleScene(sc)
    ref = osc.render_pt(oracle.options_for_scene(sc, **kw), 112, 112)
    print("draw", k, kw)
    for i, m in enumerate(mats): print("  mat", i, m)
    for ieee in (False, True):
        img = T._render_hip(R, ctx, ss, sc, (112, 112), ieee=ieee, **kw)
        d = np.abs(img[..., :3] - ref[..., :3]).reshape(-1, 112, 112, 3)[0] if img.ndim == 4 else np.abs(img[..., :3] - ref[..., :3])
        im = img.reshape(-1, 112, 112, 4)[0]; rf = ref.reshape(-1, 112, 112, 4)[0]
        d = np.abs(im[..., :3] - rf[..., :3])
        print("ieee", ieee, "mean hip", im[..., :3].mean(), "mean ref", rf[..., :3].mean(), "sum |d|", d.sum(), "max ref", rf[..., :3].max())
        idx = np.argsort(d.max(-1).ravel())[::-1][:8]
        for j in idx:
            y, x = divmod(int(j), 112)
            print("   px", x, y, "hip", im[y, x, :3], "ref", rf[y, x, :3])
