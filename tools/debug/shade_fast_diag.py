#!/usr/bin/env python3
"""Where do the two arithmetic modes of k_shade (csrc/shade_fast.hip vs IEEE, TRHIP_SHADE_FAST=0) part?  Renders the textured
quad of tests/test_gpu_parity.py::test_texture_edge_cases in both modes (this process = the mode of the environment; the other
mode in a child) and against the oracle, per bounce count.  usage (GPU box): python tools/debug/shade_fast_diag.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def scene():
    from tauray_amd import scene as S
    rng = np.random.default_rng(5)
    texs = [rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) for (w, h) in ((3, 5), (1, 1), (7, 2), (4, 4))]
    texs[0][..., 3] = rng.integers(0, 2, size=(5, 3)) * 255
    texs[2][..., 2] = 255
    quad = np.zeros(4, dtype=S.VERTEX)
    quad["pos"] = [(-1.5, -1.5, 0), (1.5, -1.5, 0), (1.5, 1.5, 0), (-1.5, 1.5, 0)]
    quad["normal"] = (0, 0, 1)
    quad["tangent"] = (1, 0, 0, 1)
    quad["uv"] = [(-2.3, -1.7), (3.7, -1.7), (3.7, 2.9), (-2.3, 2.9)]
    back = quad.copy()
    back["pos"][:, 2] = -1.0
    back["uv"] = [(0, 0), (1, 0), (1, 1), (0, 1)]
    mat = S.make_material(albedo=(0.9, 0.8, 0.7, 1.0), metallic=0.7, roughness=0.8, emission=(0.3, 0.2, 0.1), albedo_tex=0, mr_tex=1, normal_tex=2,
                          emission_tex=3, double_sided=True)
    plain = S.make_material(albedo=(0.5, 0.5, 0.9, 1.0), metallic=0.0, roughness=0.7)
    cam = S.Camera(fov=60, aspect=1.0)
    cam.transform = S.trs_matrix((0.1, -0.05, 3))
    return S.SceneDesc(instances=np.concatenate([S.make_instance(np.eye(4), mat), S.make_instance(np.eye(4), plain)]),
                       spans=np.array([(0, 4, 0, 2), (4, 4, 6, 2)], dtype=S.MESH_SPAN), vertices=np.concatenate([quad, back]),
                       indices=np.array([0, 1, 2, 0, 2, 3] * 2, dtype=np.uint32), point_lights=S.make_point_light((20, 20, 20), (0.5, 0.8, 2.5), 0.1),
                       textures=texs, cameras=[cam]).finalize(True)


def render(bounces, targets=False):
    from tauray_amd import renderer as R
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    sc = scene()
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, sc)
    W = H = 192
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(sc, max_bounces=bounces), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    buf = ctx.alloc(W * H * 16).zero()
    pt.run(buf)
    return buf.download((H, W, 4))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        np.save(sys.argv[3], render(int(sys.argv[2])))
        sys.exit(0)
    from oracle import binding as B
    sc = scene()
    osc = B.OracleScene(sc)
    for bounces in (1, 2, 3):
        fast = render(bounces)
        out = f"/tmp/ieee_{bounces}.npy"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", str(bounces), out], env=dict(os.environ, TRHIP_SHADE_FAST="0"))
        ieee = np.load(out)
        ref = osc.render_pt(B.options_for_scene(sc, max_bounces=bounces), 192, 192)[0]
        def bad(a, b):
            rel = np.abs(a[..., :3] - b[..., :3]) / (np.abs(b[..., :3]) + 1e-2)
            return float((rel.max(-1) > 1e-2).mean()), float(np.abs(a[..., :3] - b[..., :3]).max()), float(a[..., :3].mean()), float(b[..., :3].mean())
        print(f"bounces {bounces}: fast vs oracle {bad(fast, ref)}  ieee vs oracle {bad(ieee, ref)}  fast vs ieee {bad(fast, ieee)}  nan fast/ieee/ref {int(np.isnan(fast).sum())} {int(np.isnan(ieee).sum())} {int(np.isnan(ref).sum())}")
        rel = np.abs(fast[..., :3] - ieee[..., :3]) / (np.abs(ieee[..., :3]) + 1e-2)
        ys, xs = np.where(rel.max(-1) > 1e-2)
        for y, x in list(zip(ys, xs))[:6]:
            print("   pixel", x, y, "fast", fast[y, x, :3], "ieee", ieee[y, x, :3], "oracle", ref[y, x, :3])
