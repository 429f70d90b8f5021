"""Seeded procedural scenes for the benchmark configurations.

The reference repository ships no Sponza asset, so configs 3-5 of BASELINE.json
run on a generated "Sponza-class" atrium with the value distributions fixed in SURVEY.md section 8(d):
closed 30 x 12 x 14 atrium, two storeys of arcades (tessellated columns and arches), alpha-tested
curtains (~5 % of triangles), ~25 materials with roughness ~ U[0.1, 1] and 10 % metals, four procedural
1024^2 RGBA8 textures, a 0.5 degree sun through the open roof, a uniform white environment, two emissive
quads, camera at one end looking down the nave.  `sponza_teapots` adds 50 copies of the 14 280-triangle
teapot of test.glb (10 glass, 10 metal, 30 diffuse).  Output is deterministic for a given seed; the
content hash is reported with benchmark results.
"""
from __future__ import annotations

import hashlib
import math
import os

import numpy as np

from . import scene as S

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_GLB = os.path.join(_ROOT, "tests", "golden", "test.glb")


def _grid_mesh(fn, nu, nv, flip=False):
    """Tessellate a parametric surface fn(u, v) -> (pos, normal) on an nu x nv quad grid."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    pos, nrm = fn(uu.reshape(-1), vv.reshape(-1))
    verts = np.zeros(len(pos), dtype=S.VERTEX)
    verts["pos"] = pos
    verts["normal"] = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    verts["uv"] = np.stack([uu.reshape(-1), vv.reshape(-1)], axis=1)
    # tangent = d pos / du (finite difference), orthogonalised
    eps = 1e-3
    p2, _ = fn(np.clip(uu.reshape(-1) + eps, 0, 1 + eps), vv.reshape(-1))
    t = p2 - pos
    t = t - verts["normal"] * np.sum(t * verts["normal"], axis=1, keepdims=True)
    tl = np.linalg.norm(t, axis=1, keepdims=True)
    t = np.where(tl > 1e-12, t / np.maximum(tl, 1e-12), np.array([[1.0, 0, 0]]))
    verts["tangent"][:, :3] = t
    verts["tangent"][:, 3] = 1.0
    i = np.arange(nu)[None, :] + (nu + 1) * np.arange(nv)[:, None]
    a, b, c, d = i, i + 1, i + nu + 1, i + nu + 2
    tri = np.stack([a, b, d, a, d, c], axis=-1).reshape(-1, 3)
    if flip:
        tri = tri[:, ::-1]
    return verts, tri.astype(np.uint32).reshape(-1)


def _quad(p0, eu, ev, nu=1, nv=1, uv_scale=1.0):
    p0, eu, ev = (np.asarray(x, dtype=np.float64) for x in (p0, eu, ev))
    n = np.cross(eu, ev)

    def fn(u, v):
        return p0[None] + u[:, None] * eu[None] + v[:, None] * ev[None], np.broadcast_to(n, (len(u), 3)).copy()

    verts, idx = _grid_mesh(fn, nu, nv)
    verts["uv"] *= uv_scale
    return verts, idx


def _cylinder(base, radius, height, nu, nv):
    base = np.asarray(base, dtype=np.float64)

    def fn(u, v):
        a = u * 2 * math.pi
        # slight entasis + fluting so that the tessellation carries geometric detail
        r = radius * (1.0 - 0.12 * v) * (1.0 + 0.03 * np.cos(12 * a))
        n = np.stack([np.cos(a), np.zeros_like(a), np.sin(a)], axis=1)
        p = base[None] + np.stack([r * np.cos(a), v * height, r * np.sin(a)], axis=1)
        return p, n

    return _grid_mesh(fn, nu, nv, flip=True)


def _arch(center, span, tube, axis, nu, nv):
    """Half torus (an arch) in the plane spanned by `axis` (unit horizontal) and +Y."""
    center, axis = np.asarray(center, dtype=np.float64), np.asarray(axis, dtype=np.float64)
    side = np.cross(axis, [0, 1.0, 0])

    def fn(u, v):
        a = u * math.pi                    # along the arch
        b = v * 2 * math.pi                # around the tube
        ring = np.cos(a)[:, None] * axis[None] + np.sin(a)[:, None] * np.array([0, 1.0, 0])[None]
        n = np.cos(b)[:, None] * ring + np.sin(b)[:, None] * side[None]
        p = center[None] + ring * (span * 0.5) + n * tube
        return p, n

    return _grid_mesh(fn, nu, nv)


def _procedural_texture(rng, size, kind):
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / size
    base = rng.uniform(0.35, 0.9, size=3).astype(np.float32)
    n = np.zeros((size, size), dtype=np.float32)
    for o in range(5):
        f = 2 ** (o + 2)
        ph = rng.uniform(0, 2 * math.pi, size=4)
        n += (np.sin(2 * math.pi * f * x + ph[0]) * np.sin(2 * math.pi * f * y + ph[1])
              + np.sin(2 * math.pi * f * (x + y) + ph[2]) * 0.5) / (o + 1)
    n = (n - n.min()) / (n.max() - n.min())
    img = np.zeros((size, size, 4), dtype=np.float32)
    img[..., :3] = base[None, None] * (0.6 + 0.4 * n[..., None])
    img[..., 3] = 1.0
    if kind == "curtain":   # alpha-tested weave: ~35 % holes
        holes = (np.sin(2 * math.pi * 24 * x) * np.sin(2 * math.pi * 24 * y)) > 0.35
        img[..., 3] = np.where(holes, 0.0, 1.0)
    return np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)


class _Builder:
    def __init__(self):
        self.inst, self.spans, self.verts, self.idx = [], [], [], []
        self.voff = self.ioff = 0

    def add(self, verts, idx, material, model=None):
        model = np.eye(4) if model is None else model
        self.inst.append(S.make_instance(model, material))
        self.spans.append((self.voff, len(verts), self.ioff, len(idx) // 3))
        self.verts.append(verts)
        self.idx.append(idx)
        self.voff += len(verts)
        self.ioff += len(idx)

    def tri_count(self):
        return self.ioff // 3


def _load_teapot():
    from .gltf import load_glb
    d = load_glb(TEST_GLB, 64, 64)
    sp = d.spans[4]   # instance 4 = the teapot primitive (14 280 triangles)
    v = d.vertices[sp["vertex_offset"]:sp["vertex_offset"] + sp["vertex_count"]].copy()
    i = d.indices[sp["index_offset"]:sp["index_offset"] + 3 * sp["triangle_count"]].copy()
    return v, i


def sponza_class(seed: int = 1, target_tris: int = 260_000, teapots: int = 0, width: int = 1920, height: int = 1080) -> S.SceneDesc:
    rng = np.random.default_rng(seed)
    b = _Builder()
    LX, LY, LZ = 30.0, 12.0, 14.0          # length (x), height (y), width (z)
    hx, hz = LX / 2, LZ / 2

    textures = [_procedural_texture(rng, 1024, "stone"), _procedural_texture(rng, 1024, "stone"),
                _procedural_texture(rng, 1024, "stone"), _procedural_texture(rng, 1024, "curtain")]

    def rand_material(tex=-1, double_sided=False):
        metal = rng.uniform() < 0.10
        albedo = tuple(rng.uniform(0.3, 0.9, size=3)) + (1.0,)
        return S.make_material(albedo=albedo, metallic=1.0 if metal else 0.0, roughness=float(rng.uniform(0.1, 1.0)),
                               albedo_tex=tex, double_sided=double_sided)

    mats = [rand_material(tex=int(rng.integers(0, 3)) if rng.uniform() < 0.5 else -1) for _ in range(22)]

    # --- shell: floor, four walls, ceiling with an open roof slot (sun enters there)
    shell_div = 24
    b.add(*_quad([-hx, 0, -hz], [0, 0, LZ], [LX, 0, 0], shell_div, shell_div, 6.0), mats[0])            # floor (normal +y)
    b.add(*_quad([-hx, 0, -hz], [LX, 0, 0], [0, LY, 0], shell_div, shell_div, 4.0), mats[1])            # wall z = -hz (normal +z)
    b.add(*_quad([-hx, 0, hz], [0, LY, 0], [LX, 0, 0], shell_div, shell_div, 4.0), mats[2])             # wall z = +hz (normal -z)
    b.add(*_quad([-hx, 0, -hz], [0, LY, 0], [0, 0, LZ], shell_div, shell_div, 4.0), mats[3])            # wall x = -hx (normal +x)
    b.add(*_quad([hx, 0, -hz], [0, 0, LZ], [0, LY, 0], shell_div, shell_div, 4.0), mats[4])             # wall x = +hx (normal -x)
    slot = 4.0   # open strip along the nave
    b.add(*_quad([-hx, LY, -hz], [LX, 0, 0], [0, 0, hz - slot / 2], shell_div, 8, 4.0), mats[5])        # ceiling halves (normal -y)
    b.add(*_quad([-hx, LY, slot / 2], [LX, 0, 0], [0, 0, hz - slot / 2], shell_div, 8, 4.0), mats[5])
    # upper gallery floors along both sides
    gal_w = 3.0
    for sgn in (-1, 1):
        z0 = sgn * hz - (gal_w if sgn > 0 else 0)
        b.add(*_quad([-hx, LY / 2, z0], [0, 0, gal_w], [LX, 0, 0], shell_div, 4, 4.0), mats[6], None)
        b.add(*_quad([-hx, LY / 2 - 0.3, z0], [LX, 0, 0], [0, 0, gal_w], shell_div, 4, 4.0), mats[6], None)

    # --- arcades: columns + arches, two storeys, both sides.  Tessellation is derived from the budget.
    n_cols = 11
    fixed = b.tri_count()
    teapot_tris = 14280 * teapots
    n_curtains = 8
    curtain_tris_target = int(0.05 * target_tris)
    cur_div = max(2, int(math.sqrt(curtain_tris_target / n_curtains / 2)))
    budget = max(target_tris - teapot_tris - fixed - n_curtains * 2 * cur_div * cur_div, 20000)
    n_col_total = n_cols * 2 * 2
    n_arch_total = (n_cols - 1) * 2 * 2
    per_obj = budget / (n_col_total + n_arch_total)
    cu = max(8, int(math.sqrt(per_obj / 2 * 2)))       # around
    cv = max(4, int(per_obj / 2 / cu))                 # along
    col_x = np.linspace(-hx + 1.5, hx - 1.5, n_cols)
    for storey in range(2):
        y0 = storey * LY / 2
        col_h = LY / 2 - 1.6
        for sgn in (-1, 1):
            z = sgn * (hz - gal_w)
            for k, x in enumerate(col_x):
                b.add(*_cylinder([x, y0, z], 0.35, col_h, cu, cv), mats[7 + (k + storey) % 6])
            for k in range(n_cols - 1):
                xc = 0.5 * (col_x[k] + col_x[k + 1])
                span = col_x[k + 1] - col_x[k]
                b.add(*_arch([xc, y0 + col_h, z], span, 0.22, [1.0, 0, 0], cu, cv), mats[13 + (k + storey) % 5])

    # --- curtains: alpha-tested quads hanging in the upper arcade openings
    cur_mat = S.make_material(albedo=(0.8, 0.25, 0.2, 1.0), metallic=0.0, roughness=0.8, albedo_tex=3, double_sided=True)
    for k in range(n_curtains):
        sgn = -1 if k % 2 == 0 else 1
        x = col_x[1 + k] if 1 + k < n_cols - 1 else col_x[k % (n_cols - 1)]
        z = sgn * (hz - gal_w) - sgn * 0.05
        wv = col_x[1] - col_x[0] - 0.8
        b.add(*_quad([x + 0.4, LY / 2 + 0.2, z], [wv, 0, 0], [0, LY / 2 - 2.2, 0], cur_div, cur_div, 2.0), cur_mat)

    # --- two emissive quads (lanterns)
    em_mat = S.make_material(albedo=(0, 0, 0, 1), metallic=0.0, roughness=1.0, emission=(12.0, 9.0, 5.0))
    b.add(*_quad([-6.0, LY / 2 - 0.6, -0.75], [1.5, 0, 0], [0, 0, 1.5], 1, 1), em_mat)
    b.add(*_quad([6.0, LY / 2 - 0.6, -0.75], [1.5, 0, 0], [0, 0, 1.5], 1, 1), em_mat)

    # --- teapots
    if teapots:
        tv, ti = _load_teapot()
        for k in range(teapots):
            if k < 10:
                m = S.make_material(albedo=tuple(rng.uniform(0.6, 1.0, 3)) + (1.0,), metallic=0.0, roughness=0.05,
                                    transmittance=1.0, ior=1.45, double_sided=True)
            elif k < 20:
                m = S.make_material(albedo=tuple(rng.uniform(0.5, 0.95, 3)) + (1.0,), metallic=1.0, roughness=float(rng.uniform(0.1, 0.4)))
            else:
                m = S.make_material(albedo=tuple(rng.uniform(0.2, 0.9, 3)) + (1.0,), metallic=0.0, roughness=float(rng.uniform(0.3, 1.0)))
            pos = [rng.uniform(-hx + 2, hx - 2), 0.0 if rng.uniform() < 0.7 else LY / 2, rng.uniform(-hz + gal_w + 0.8, hz - gal_w - 0.8)]
            if pos[1] > 0:
                pos[2] = math.copysign(hz - gal_w / 2, rng.uniform(-1, 1))
            ang = rng.uniform(0, 2 * math.pi)
            q = (0.0, math.sin(ang / 2), 0.0, math.cos(ang / 2))
            sc = float(rng.uniform(0.25, 0.5))
            b.add(tv, ti, m, S.trs_matrix(pos, q, (sc, sc, sc)))

    cam = S.Camera(projection=S.PROJ_PERSPECTIVE, fov=60.0, aspect=width / float(height), near=0.1, far=200.0)
    # camera at the -x end, 1.7 m above the floor, looking down the nave (+x): -Z of the camera maps to +X
    rot = np.array([[0, 0, -1.0], [0, 1, 0], [1.0, 0, 0]])
    t = np.eye(4)
    t[:3, :3] = rot
    t[:3, 3] = [-hx + 1.0, 1.7, 0.0]
    cam.transform = t

    sun_dir = np.array([0.25, -1.0, 0.18])
    desc = S.SceneDesc(
        instances=np.concatenate(b.inst), spans=np.array(b.spans, dtype=S.MESH_SPAN), vertices=np.concatenate(b.verts),
        indices=np.concatenate(b.idx).astype(np.uint32),
        point_lights=np.zeros(0, dtype=S.POINT_LIGHT),
        directional_lights=S.make_directional_light((6.0, 5.6, 5.0), sun_dir, 0.5),
        textures=textures, envmap=np.ones((4, 8, 4), dtype=np.float32), environment_factor=(1.0, 1.0, 1.0, 1.0),
        cameras=[cam], name=f"sponza_class(seed={seed}, teapots={teapots})")
    return desc.finalize(True)


def sponza_teapots(seed: int = 1, width: int = 1920, height: int = 1080) -> S.SceneDesc:
    """~1.0 M triangles: the atrium at ~286k + 50 teapots x 14 280."""
    return sponza_class(seed=seed, target_tris=1_000_000, teapots=50, width=width, height=height)


def scene_hash(desc: S.SceneDesc) -> str:
    h = hashlib.sha256()
    for a in (desc.instances, desc.spans, desc.vertices, desc.indices, desc.directional_lights, desc.point_lights):
        h.update(np.ascontiguousarray(a).tobytes())
    for t in desc.textures:
        h.update(np.ascontiguousarray(t).tobytes())
    return h.hexdigest()[:16]


def test_glb(width: int = 1920, height: int = 1080) -> S.SceneDesc:
    from .gltf import load_glb
    return load_glb(TEST_GLB, width, height)


WORKLOADS = {
    "test_glb": test_glb,
    "sponza_class": lambda width=1920, height=1080: sponza_class(1, 260_000, 0, width, height),
    "sponza_teapots": lambda width=1920, height=1080: sponza_teapots(1, width, height),
}
