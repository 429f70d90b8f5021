"""Known-answer tests of the oracle's pure functions against independent numpy restatements
(SURVEY.md Appendix D item 3): RNG hashes, Sobol/Owen machinery, packing, distribution math, camera,
sampling routines and tonemap."""
import ctypes as C
import math

import numpy as np
import pytest

U32 = np.uint32


def _pcg_np(seed):
    with np.errstate(over="ignore"):
        seed = U32(seed) * U32(747796405) + U32(2891336453)
        seed = ((seed >> ((seed >> U32(28)) + U32(4))) ^ seed) * U32(277803737)
        return (seed >> U32(22)) ^ seed


def _pcg4d_np(v):
    with np.errstate(over="ignore"):
        v = np.array(v, dtype=U32) * U32(1664525) + U32(1013904223)
        def mixin(v):
            return v + v[[1, 2, 0, 1]] * v[[3, 0, 1, 2]]
        v = mixin(v)
        v = (v >> U32(16)) ^ v
        v = mixin(v)
    return v


def _u32arr(*v):
    return (C.c_uint32 * len(v))(*[int(x) for x in v])


def test_pcg_family(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    for s in rng.integers(0, 2**32, size=200, dtype=np.uint64):
        seed = C.c_uint32(int(s))
        r = L.oracle_pcg(C.byref(seed))
        assert r == int(_pcg_np(s)) and seed.value == r       # inout: the argument is mutated to the result
    assert L.oracle_pcg(C.byref(C.c_uint32(0))) == int(_pcg_np(0))
    for _ in range(100):
        v = rng.integers(0, 2**32, size=4, dtype=np.uint64)
        seed, out = _u32arr(*v), _u32arr(0, 0, 0, 0)
        L.oracle_pcg4d(seed, out)
        exp = _pcg4d_np(v)
        assert list(out) == [int(x) for x in exp] and list(seed) == list(out)


def test_init_random_sampler(oracle):
    L = oracle.lib()
    coord = np.array([17, 3999, 2, 123456], dtype=np.uint64)
    out = _u32arr(0, 0, 0, 0)
    L.oracle_init_random_sampler(_u32arr(*coord), out)
    s = [int(c) for c in coord]
    s[0] = int(_pcg_np(s[0])); s[1] ^= s[0]
    s[1] = int(_pcg_np(s[1])); s[2] ^= s[1]
    s[2] = int(_pcg_np(s[2])); s[3] ^= s[2]
    # pcg mutates x, y, z in place while xoring into the next component (random_sampler.glsl:15-17)
    assert list(out) == s


def _bitrev(x):
    return int(format(int(x) & 0xFFFFFFFF, "032b")[::-1], 2)


def test_morton_and_permutation(oracle):
    L = oracle.lib()
    def m2(x, y):
        r = 0
        for b in range(16):
            r |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
        return r
    def m3(x, y, z):
        r = 0
        for b in range(10):
            r |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
        return r
    for x, y, z in [(0, 0, 0), (1, 2, 3), (1919, 1079, 511), (65535, 65535, 1023), (70000, 5, 4096)]:
        assert L.oracle_morton_2d(x, y) == m2(x & 0xFFFF, y & 0xFFFF)
        assert L.oracle_morton_3d(x, y, z) == m3(x & 1023, y & 1023, z & 1023)
    # get_permutation_n(n, p, .) enumerates a permutation of 0..n-1 for every p
    for n, count in ((4, 24), (8, 40320)):
        seen = set()
        for p in list(range(24)) + [count - 1, count // 2]:
            perm = tuple(L.oracle_get_permutation_n(n, p % count, d) for d in range(n))
            assert sorted(perm) == list(range(n))
            seen.add(perm)
        assert len(seen) >= 24


def test_owen_scramble_2d_is_lk_hash(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    M = 0xFFFFFFFF
    for _ in range(50):
        x = [int(v) for v in rng.integers(0, 2**32, size=4, dtype=np.uint64)]
        s = [int(v) for v in rng.integers(0, 2**32, size=4, dtype=np.uint64)]
        out = _u32arr(0, 0, 0, 0)
        L.oracle_owen_scramble_2d(_u32arr(*x), _u32arr(*s), out)
        for k in range(4):
            v = _bitrev(x[k])
            v ^= (v * 0x3D20ADEA) & M
            v = (v + s[k]) & M
            v = (v * ((s[k] >> 16) | 1)) & M
            v ^= (v * 0x05526C56) & M
            v ^= (v * 0x53A22864) & M
            assert out[k] == _bitrev(v)


def test_sobol_sample_uses_joe_kuo_numbers_and_keeps_reference_quirks(oracle):
    import os
    from conftest import GOLDEN
    L = oracle.lib()
    table = np.load(os.path.join(GOLDEN, "sobol_table.npy"))
    out = _u32arr(0, 0, 0, 0)
    for index in [0, 1, 2, 3, 5, 0b1011, 12345, 0x80000001, 0xFFFFFFFF]:
        for bounce in range(4):
            L.oracle_generate_sobol_sample(index, bounce, 4, out)
            exp = np.zeros(4, dtype=np.uint32)
            if index:
                lsb = (index & -index).bit_length() - 1
                msb = index.bit_length() - 1
                for bit in range(lsb, msb):            # the top set bit is excluded (math.glsl:137-142)
                    if (index >> bit) & 1:
                        exp ^= table[bounce * 32 + bit]
            assert list(out) == [int(v) for v in exp], (index, bounce)
    # dimension sets >= MAX_SOBOL_BOUNCES fall back to pcg4d(index, bounce, bounce*index, 0)
    L.oracle_generate_sobol_sample(77, 5, 4, out)
    assert list(out) == [int(v) for v in _pcg4d_np([77, 5, (5 * 77) & 0xFFFFFFFF, 0])]


@pytest.mark.parametrize("sampler", [0, 1, 2, 3])
def test_ray_sample_streams_are_deterministic_and_distinct(oracle, sampler):
    L = oracle.lib()
    def draw(coord, bounce, sc=0, seed=0):
        out = _u32arr(0, 0, 0, 0)
        L.oracle_ray_sample_uint(sampler, 4, _u32arr(*coord), sc, seed, bounce, out)
        return tuple(out)
    a = draw((10, 20, 0, 0), 0)
    assert a == draw((10, 20, 0, 0), 0)
    assert a != draw((11, 20, 0, 0), 0)
    assert a != draw((10, 20, 0, 1), 0)
    if sampler != 0:   # the uniform-random stream is stateful: bounce_index is ignored
        assert a != draw((10, 20, 0, 0), 1)
    assert draw((10, 20, 0, 0), 0, sc=5) == draw((10, 20, 0, 5), 0)      # coord.w += sample_counter
    if sampler in (0, 1):
        assert a != draw((10, 20, 0, 0), 0, seed=7)                      # coord.z += pcg(rng_seed)
    # uniformity smoke check over pixels
    vals = np.array([draw((x, y, 0, 0), 0) for x in range(24) for y in range(24)], dtype=np.float64) / 2**32
    assert abs(vals.mean() - 0.5) < 0.05


def test_packing(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(5)
    for _ in range(200):
        x, y = rng.normal(size=2) * rng.choice([1e-6, 1e-3, 1, 100, 7e4])
        exp = int(np.float16(np.float32(x)).view(np.uint16)) | (int(np.float16(np.float32(y)).view(np.uint16)) << 16)
        assert L.oracle_pack_half2x16(float(np.float32(x)), float(np.float32(y))) == exp
    for rgb in [(1, 1, 0.17924630641937256), (12, 9, 5), (0.001, 0.5, 0.25), (0, 0, 0), (70000, 1, 1)]:
        v = L.oracle_rgb_to_r9g9b9e5((C.c_float * 3)(*rgb))
        back = (C.c_float * 3)()
        L.oracle_r9g9b9e5_to_rgb(v, back)
        m = max(rgb)
        if 0 < m < 32768:
            assert np.allclose(list(back), rgb, atol=m / 256 + 1e-9)


def test_distribution_math_matches_host_mirror(oracle):
    from tauray_amd import distribution as D
    L = oracle.lib()
    for size in [(1920, 1080), (512, 512), (33, 7), (1, 1)]:
        b = D.calculate_shuffled_strips_b(size)
        n = size[0] * size[1]
        assert b == 0 or (n >> b) >= 128
        region = ((n + (1 << b) - 1) >> b)
        idx = np.arange(region << b)
        perm = np.array([L.oracle_permute_region_id(int(i), size[0], size[1], b) for i in idx[:: max(1, len(idx) // 4000)]])
        host = np.array([D.permute_region_id(int(i), size, b) for i in idx[:: max(1, len(idx) // 4000)]])
        assert np.array_equal(perm, host)
    # permutation of the padded index space is a bijection
    size, b = (64, 48), D.calculate_shuffled_strips_b((64, 48))
    region = ((64 * 48 + (1 << b) - 1) >> b)
    full = sorted(D.permute_region_id(i, size, b) for i in range(region << b))
    assert full == list(range(region << b))
    # scanline shares tile the image
    for count in (2, 3, 8):
        rows = sum(D.get_distribution_render_size(D.DistributionParams((1920, 1080), D.DISTRIBUTION_SCANLINE, i, count, i == 0))[1] for i in range(count))
        assert rows == 1080
    # shuffled strips: device pixel ranges tile [0, region << b)
    size = (1920, 1080)
    ratios = [0.1, 0.2, 0.3, 0.4]
    cum, total, prev_end = 0.0, 0, 0
    for i, r in enumerate(ratios):
        p = D.get_device_distribution_params(size, D.DISTRIBUTION_SHUFFLED_STRIPS, cum, r, i, 4, i == 0)
        assert p.index == prev_end
        prev_end = p.index + p.count
        cum += r
    b = D.calculate_shuffled_strips_b(size)
    assert prev_end >= size[0] * size[1] and prev_end <= (D.get_region_size(size[0] * size[1], b) << b) + 4


def test_camera_ray_perspective(oracle, test_glb_512):
    L = oracle.lib()
    cam = test_glb_512.cameras[0]
    data = cam.pack()
    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    t = math.tan(math.radians(cam.fov) / 2)
    for px, py in [(0.5, 0.5), (256.0, 256.0), (511.5, 100.25)]:
        L.oracle_camera_ray(data.ctypes.data, 0, px, py, 512.0, 512.0, 0.5, 0.5, 0, o, d)
        u, v = px / 512 * 2 - 1, py / 512 * 2 - 1
        exp = np.array([u * t * cam.aspect, v * t, -1.0])
        exp = cam.transform[:3, :3] @ (exp / np.linalg.norm(exp))
        assert np.allclose(list(d), exp, atol=2e-6)
        assert np.allclose(list(o), cam.transform[:3, 3], atol=1e-6)


def test_sampling_routines(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(9)
    out = (C.c_float * 3)()
    for _ in range(200):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        cmin = float(rng.uniform(-0.5, 0.9999))
        L.oracle_sample_cone(float(rng.uniform()), float(rng.uniform()), (C.c_float * 3)(*d), cmin, out)
        o = np.array(list(out))
        assert abs(np.linalg.norm(o) - 1) < 1e-4 and o @ d >= cmin - 1e-5
    pdf = C.c_float()
    for _ in range(100):
        A, B, Cc = (rng.normal(size=3) + np.array([0, 0, 3.0]) for _ in range(3))
        L.oracle_sample_spherical_triangle(float(rng.uniform()), float(rng.uniform()), (C.c_float * 3)(*A), (C.c_float * 3)(*B),
                                           (C.c_float * 3)(*Cc), out, C.byref(pdf))
        o = np.array(list(out))
        a, b, c = (v / np.linalg.norm(v) for v in (A, B, Cc))
        omega = 2 * math.atan2(abs(np.dot(a, np.cross(b, c))), 1 + a @ b + b @ c + a @ c)
        assert abs(1 / pdf.value - omega) < 2e-3 * max(omega, 1e-3)
        # the direction lies inside the spherical triangle: same side of the three edge planes
        s = np.sign(np.dot(a, np.cross(b, c)))
        assert all(s * np.dot(o, np.cross(p, q)) > -1e-3 for p, q in ((a, b), (b, c), (c, a)))
    assert L.oracle_sample_blackman_harris(0.5) == pytest.approx(0.5, abs=2e-3)
    xs = [L.oracle_sample_blackman_harris(float(u)) for u in np.linspace(0.01, 0.99, 50)]
    assert all(b >= a - 1e-6 for a, b in zip(xs, xs[1:]))


def test_ggx_sample_and_pdf_agree(oracle):
    """material_bsdf_sample's pdf and lobes equal what material_bsdf_pdf evaluates for the sampled direction."""
    L = oracle.lib()
    rng = np.random.default_rng(11)
    mats = [(0.8, 0.5, 0.3, 1, 0.0, 0.25, 0.0, 1.0, 1.45), (0.9, 0.9, 0.9, 1, 1.0, 0.04, 0.0, 1.0, 1.45),
            (0.5, 0.6, 0.3, 1, 0.0, 0.04, 1.0, 1.0, 1.45), (0.5, 0.6, 0.3, 1, 0.0, 0.09, 1.0, 1.45, 1.0)]
    checked = 0
    for m in mats:
        f0 = ((m[8] - m[7]) / (m[8] + m[7])) ** 2
        mat = (C.c_float * 10)(*m, f0)
        for _ in range(300):
            v = rng.normal(size=3); v[2] = abs(v[2]) + 0.05; v /= np.linalg.norm(v)
            u = (C.c_float * 4)(*rng.uniform(size=4))
            od, lobes, pdf = (C.c_float * 3)(), (C.c_float * 4)(), C.c_float()
            L.oracle_ggx_bsdf_sample(u, (C.c_float * 3)(*v), mat, od, lobes, C.byref(pdf))
            if pdf.value <= 0 or not np.isfinite(pdf.value):
                continue
            if od[2] <= 0 and m[6] == 0:
                # a VNDF reflection can land below the horizon: the sample keeps its pdf but every lobe is 0
                assert all(x == 0 for x in lobes)
                continue
            lobes2 = (C.c_float * 4)()
            pdf2 = L.oracle_ggx_bsdf_pdf(od, (C.c_float * 3)(*v), mat, lobes2)
            if m[6] > 0:
                # transmissive lobe: the reference's sampling pdf and evaluation pdf differ by design
                # (ggx.glsl:360 vs :489 carry a factor pi and different masking terms): only sanity-check
                assert np.isfinite(pdf2) and pdf2 >= 0 and all(np.isfinite(list(lobes)))
                continue
            assert pdf2 == pytest.approx(pdf.value, rel=2e-3, abs=1e-6)
            assert np.allclose(list(lobes), list(lobes2), rtol=5e-3, atol=1e-5)
            checked += 1
    assert checked > 300


def test_filmic_tonemap(oracle):
    x = np.array([[0, 0.004, 0.18, 1.0], [2.0, 10.0, 1e4, 1.0], [1.0, 1.0, 0.17924630641937256, 0.5]], dtype=np.float32)
    out = oracle.tonemap(x.reshape(1, 3, 4), op=2, exposure=1.0, gamma=2.2).reshape(3, 4)
    c = np.maximum(np.clip(x[:, :3].astype(np.float64), 0, 1000) - 0.004, 0)
    exp = (c * (6.2 * c + 0.5)) / (c * (6.2 * c + 1.7) + 0.06)     # pow(.,2.2) then pow(.,1/2.2)
    assert np.allclose(out[:, :3], exp, atol=2e-6)
    assert np.array_equal(out[:, 3], x[:, 3])
    lin = oracle.tonemap(x.reshape(1, 3, 4), op=0, exposure=2.0, gamma=2.2).reshape(3, 4)
    assert np.allclose(lin[:, :3], x[:, :3] * 2.0)


def _random_affine(rng, scale_range=(0.5, 1.5)):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    m = np.eye(4)
    m[:3, :3] = q @ np.diag(rng.uniform(*scale_range, size=3))
    m[:3, 3] = rng.uniform(-2, 2, size=3)
    return m


def test_skinning_against_float64(oracle):
    """shader/skinning.comp: skin_mat = sum w_k joint_k, positions through it, normals and tangents through its inverse
    transpose and renormalised; checked against a float64 numpy evaluation."""
    from tauray_amd.scene import VERTEX, SKIN
    rng = np.random.default_rng(7)
    n, nj = 500, 6
    src = np.zeros(n, dtype=VERTEX)
    src["pos"] = rng.uniform(-3, 3, size=(n, 3))
    nrm = rng.normal(size=(n, 3)); src["normal"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    tan = rng.normal(size=(n, 3)); src["tangent"][:, :3] = tan / np.linalg.norm(tan, axis=1, keepdims=True)
    src["tangent"][:, 3] = rng.choice([-1.0, 1.0], size=n)
    src["uv"] = rng.uniform(size=(n, 2))
    skins = np.zeros(n, dtype=SKIN)
    skins["joints"] = rng.integers(0, nj, size=(n, 4))
    w = rng.uniform(size=(n, 4)); w[rng.uniform(size=(n, 4)) < 0.3] = 0.0; w[:, 0] += 1e-3
    skins["weights"] = w / w.sum(axis=1, keepdims=True)
    joints = np.stack([_random_affine(rng) for _ in range(nj)]).astype(np.float32)

    out = oracle.skin_vertices(src, skins, joints)
    j64 = joints.astype(np.float64)
    w64 = skins["weights"].astype(np.float64)
    for i in range(n):
        m = sum(w64[i, k] * j64[skins["joints"][i, k]] for k in range(4))
        it = np.linalg.inv(m).T
        p = m @ np.append(src["pos"][i].astype(np.float64), 1.0)
        assert np.allclose(out["pos"][i], p[:3], rtol=2e-6, atol=2e-6)
        for got, d in ((out["normal"][i], src["normal"][i]), (out["tangent"][i][:3], src["tangent"][i][:3])):
            e = it[:3, :3] @ d.astype(np.float64)
            assert np.allclose(got, e / np.linalg.norm(e), atol=5e-6)
    assert np.array_equal(out["uv"], src["uv"]) and np.array_equal(out["tangent"][:, 3], src["tangent"][:, 3])

    # identity joints: positions come back bit for bit, unit directions within an ulp of themselves
    ident = np.stack([np.eye(4, dtype=np.float32)] * nj)
    same = oracle.skin_vertices(src, skins, ident)
    one_hot = np.zeros(n, dtype=SKIN); one_hot["weights"][:, 0] = 1.0
    same1 = oracle.skin_vertices(src, one_hot, ident)
    assert np.array_equal(same1["pos"], src["pos"])
    assert np.allclose(same["pos"], src["pos"], rtol=3e-7, atol=1e-6) and np.allclose(same["normal"], src["normal"], atol=3e-7)
    # a joint id past the array falls back to joint 0 (undefined in the shader; defined here so that nothing is read out of bounds)
    wild = one_hot.copy(); wild["joints"][:, 0] = 1000
    assert np.array_equal(oracle.skin_vertices(src, wild, joints).view(np.uint8), oracle.skin_vertices(src, one_hot, joints).view(np.uint8))


def test_skinned_glb_loader(oracle):
    """tests/golden/skinned.glb (tools/make_skinned_glb.py): skins, renormalised weights, the skinned mesh moved to the
    origin, joint transforms = global transform * inverse bind matrix - recomputed here from the generator's parameters."""
    import os
    from conftest import GOLDEN
    from tauray_amd.gltf import load_glb
    scene = load_glb(os.path.join(GOLDEN, "skinned.glb"), 128, 128)
    assert len(scene.skinned) == 1 and scene.skinned[0].instance == 0 and scene.skinned[0].joint_nodes == [2, 3, 4]
    sk = scene.skinned[0]
    assert np.array_equal(np.asarray(scene.instances["model"][0]), np.eye(4, dtype=np.float32)), "skinned meshes sit at the origin"
    assert np.allclose(sk.skins["weights"].sum(axis=1), 1.0, atol=1e-6) and len(sk.skins) == scene.spans["vertex_count"][0]

    def rz(deg):
        c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
        return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])

    def tr(x, y, z):
        m = np.eye(4); m[:3, 3] = (x, y, z); return m

    g0 = tr(0.3, 0.0, -0.2) @ tr(0, 0, 0) @ rz(0.0)
    g1 = g0 @ tr(0, 1, 0) @ rz(30.0)
    g2 = g1 @ tr(0, 1, 0) @ rz(40.0)
    want = np.stack([g0 @ tr(0, 0, 0), g1 @ tr(0, -1, 0), g2 @ tr(0, -2, 0)])
    got = scene.joint_transforms(sk)
    assert np.allclose(got, want, atol=1e-6)
    # the top ring hangs on joint 2 alone: a bind-pose vertex (x, 2, z) goes to g2 * (x, 0, z)
    lo = int(scene.spans["vertex_offset"][0]); n = int(scene.spans["vertex_count"][0])
    bind = scene.vertices[lo:lo + n]
    posed = oracle.skin_vertices(bind, sk.skins, got)
    top = np.where(bind["pos"][:, 1] == 2.0)[0]
    assert len(top) == 24
    for i in top:
        e = g2 @ np.array([bind["pos"][i][0], 0.0, bind["pos"][i][2], 1.0])
        assert np.allclose(posed["pos"][i], e[:3], atol=2e-6)
    # and the bottom ring stays where the skeleton's root puts it
    bottom = np.where(bind["pos"][:, 1] == 0.0)[0]
    assert np.allclose(posed["pos"][bottom], bind["pos"][bottom] + np.array([0.3, 0.0, -0.2], np.float32), atol=1e-6)
    # the oracle scene is built from the posed vertices
    osc = oracle.OracleScene(scene)
    ids = osc.render_feature(9, 64, 64)[..., 0]
    assert (ids == 0).sum() > 20 and (ids == 1).sum() > 100


# ---------------------------------------------------------------------------------------------------
# glTF animation clips (tauray_amd/animation.py = src/animation.{hh,cc,tcc} + src/gltf.cc:167-190,580-627)
def _load_animated():
    import importlib.util
    import os
    from conftest import GOLDEN, ROOT
    from tauray_amd.gltf import load_glb
    spec = importlib.util.spec_from_file_location("make_animated_glb", os.path.join(ROOT, "tools", "make_animated_glb.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return load_glb(os.path.join(GOLDEN, "animated.glb"), 64, 64), gen


def test_animation_tracks_of_the_animated_fixture():
    """The loader turns every channel into the track its target path names, with microsecond ticks, and the interpolation rules
    of animation::interpolate: clamped outside the keys, LINEAR mix, slerp for rotations, STEP, CUBICSPLINE with tangents scaled by
    the key interval in seconds - against values recomputed here from the generator's tables."""
    from tauray_amd import animation as A
    scene, gen = _load_animated()
    names = {n: sorted(pool) for n, pool in scene.animations.items()}
    box, child = 8, 9
    assert names[box] == ["move", "spin"] and names[child] == ["move"] and sorted(names) == [3, 4, 6, 7, 8, 9]
    move = scene.animations[box]["move"]
    assert move.position.timestamps == [0, 350000, 800000, 1250000] and move.position.interpolation == A.LINEAR
    assert move.scaling.interpolation == A.STEP and move.orientation.interpolation == A.CUBICSPLINE
    assert move.loop_time == 1250000 and scene.animations[box]["spin"].loop_time == 500000
    # LINEAR: between keys 1 and 2 at t = 0.5 s, ratio (0.5 - 0.35) / 0.45
    r = np.float32(500000 - 350000) / np.float32(800000 - 350000)
    want = np.array(gen.BOX_T_VALUES[1], np.float32) * (1 - float(r)) + np.array(gen.BOX_T_VALUES[2], np.float32) * float(r)
    assert np.allclose(move.position.sample(500000), want, rtol=0, atol=1e-7)
    assert np.array_equal(move.position.sample(-5), np.array(gen.BOX_T_VALUES[0], np.float32)) and np.array_equal(move.position.sample(9_000_000), np.array(gen.BOX_T_VALUES[-1], np.float32))
    # STEP holds the earlier key; exactly on a key the later segment starts (upper_bound)
    assert np.allclose(move.scaling.sample(499999), gen.BOX_S_VALUES[0]) and np.allclose(move.scaling.sample(500000), gen.BOX_S_VALUES[1])
    # slerp: half way between 0 and 170 degrees about x is 85 degrees
    q = scene.animations[child]["move"].orientation.sample(625000, quaternion=True)
    assert np.allclose(q, gen.quat((1, 0, 0), 85.0), atol=1e-6)
    # CUBICSPLINE: Hermite form with the tangents of the file times the interval; at the keys it returns the keys
    tr = move.orientation
    assert np.allclose(tr.sample(600000, quaternion=True), gen.quat((0, 1, 0), 100.0), atol=1e-6)
    t, dt = 0.25 / 0.6, 0.6
    p1, p2 = np.array(gen.quat((0, 1, 0), 0.0)), np.array(gen.quat((0, 1, 0), 100.0))
    m1, m2 = np.array([0.0, 0.45, 0.0, -0.4]) * 0.7 * dt, np.array([0.0, 0.9, 0.0, -0.4]) * dt       # out-tangent of key 0, in-tangent of key 1
    h = (2 * t**3 - 3 * t**2 + 1) * p1 + (t**3 - 2 * t**2 + t) * m1 + (-2 * t**3 + 3 * t**2) * p2 + (t**3 - t**2) * m2
    assert np.allclose(tr.sample(250000, quaternion=True), h, atol=1e-6)


def test_animation_playback_moves_instances_cameras_and_joints():
    """SceneAnimator = play() + update() of src/scene.cc: fallback picks the alphabetically first clip per node, a named clip
    plays only where it exists, a clip that does not loop stops after its last key and leaves the pose of the frame before, a
    looping one wraps; children inherit their parents' motion; model_prev is last frame's model."""
    from tauray_amd.animation import SceneAnimator
    from tauray_amd.scene import from_glm, trs_matrix
    scene, gen = _load_animated()
    box_inst, child_inst = scene.nodes[8].instances[0], scene.nodes[9].instances[0]
    rest = scene.instances.copy()
    rest_lights = scene.point_lights.copy()
    an = SceneAnimator(scene)
    assert not an.is_playing()
    an.play("")
    assert an.is_playing() and all(c.current is scene.animations[n]["move"] for n, c in an.controllers.items())
    dt = 250000
    inst, cams, globs = an.update(0)
    assert np.allclose(from_glm(inst["model"][box_inst])[:3, 3], gen.BOX_T_VALUES[0]) and np.array_equal(inst["model_prev"], rest["model"])
    before = inst["model"].copy()
    inst, cams, globs = an.update(dt)
    assert np.array_equal(inst["model_prev"], before)
    # the box at 0.25 s: translation by LINEAR, scale by STEP (first key), rotation by the spline; its child rides on it
    m = scene.animations[8]["move"]
    want = trs_matrix(m.position.sample(dt), m.orientation.sample(dt, True) / np.linalg.norm(m.orientation.sample(dt, True)), m.scaling.sample(dt))
    assert np.allclose(from_glm(inst["model"][box_inst]), want, atol=1e-6)
    child_local = trs_matrix((0.0, 1.8, 0.0), scene.animations[9]["move"].orientation.sample(dt, True), (0.4, 0.4, 0.4))
    assert np.allclose(from_glm(inst["model"][child_inst]), want @ child_local, atol=1e-6)
    assert np.allclose(np.asarray(cams[0].transform)[:3, 3], np.array(gen.CAM_VALUES[0]) * 0.8 + np.array(gen.CAM_VALUES[1]) * 0.2, atol=1e-6)
    assert np.allclose(np.asarray(an.previous_cameras[0].transform)[:3, 3], gen.CAM_VALUES[0])
    # the lamp: a point light on an animated node, 0.25 s into its first segment (0 .. 0.6 s)
    r = float(np.float32(250000) / np.float32(600000))
    assert np.allclose(scene.point_lights["pos"][0], np.array(gen.LAMP_VALUES[0]) * (1 - r) + np.array(gen.LAMP_VALUES[1]) * r, atol=1e-6)
    assert np.array_equal(scene.point_lights["color"], rest_lights["color"])
    # joints: node 3 turns about z by the LINEAR track, the skin's joint matrices follow through node_globals
    j1 = scene.animations[3]["move"].orientation.sample(dt, True)
    assert np.allclose(globs[3], globs[2] @ trs_matrix((0, 1, 0), j1), atol=1e-6)
    assert not np.allclose(scene.joint_transforms(scene.skinned[0], globs), scene.joint_transforms(scene.skinned[0], {**globs, 3: globs[2]}))
    # the end: at 1.25 s the timer reaches the loop time, the controllers stop and the pose stays what it was at 1.0 s
    for _ in range(3):
        inst, _, _ = an.update(dt)
    assert an.is_playing()
    last = inst["model"].copy()
    inst, _, _ = an.update(dt)
    assert not an.is_playing() and np.array_equal(inst["model"], last)
    # a named clip: only the box has "spin"; looping wraps the half-second clip
    scene2, _ = _load_animated()
    an = SceneAnimator(scene2)
    an.play("spin", loop=True)
    assert [n for n, c in an.controllers.items() if c.playing] == [8]
    an.update(0)
    inst, _, _ = an.update(750000)          # 0.75 s into a 0.5 s loop = 0.25 s: 45 degrees about z
    assert an.is_playing() and np.allclose(from_glm(inst["model"][box_inst])[:3, :3] / 0.5, trs_matrix(rotation=gen.quat((0, 0, 1), 45.0))[:3, :3], atol=1e-6)


def test_hdr_reader_known_pixels(tmp_path):
    """tauray_amd/hdr.py against hand-made RGBE bytes: value = mantissa * 2^(exponent - 136) (stb_image's conversion, no half
    step), exponent 0 is black, alpha 1 is appended, row 0 is the first scanline of the file; run-length-encoded and flat
    scanlines decode to the same pixels."""
    import os
    from conftest import GOLDEN
    from tauray_amd.hdr import load_hdr
    px = [(128, 64, 32, 129), (255, 0, 1, 136), (1, 2, 3, 0), (200, 100, 50, 120)] * 2 + [(9, 9, 9, 128)]
    w = len(px)
    flat = bytes(v for p in px for v in p)
    rle = bytes([2, 2, 0, w])
    for c in range(4):
        col = [p[c] for p in px]
        rle += bytes([w]) + bytes(col)                      # one literal run per channel
    head = b"#?RADIANCE\n# made by hand\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X %d\n" % w
    f = tmp_path / "t.hdr"
    f.write_bytes(head + rle + bytes([2, 2, 0, w]) + b"".join(bytes([128 + w, v]) for v in (7, 7, 7, 130)))      # second row: four runs of one value
    img = load_hdr(str(f))
    assert img.shape == (2, w, 4) and img.dtype == np.float32
    assert np.array_equal(img[0, 0], [1.0, 0.5, 0.25, 1.0]) and np.array_equal(img[0, 1], [255.0, 0.0, 1.0, 1.0])
    assert np.array_equal(img[0, 2], [0, 0, 0, 1]) and np.array_equal(img[0, 3, :3], np.array([200, 100, 50], np.float32) * np.float32(2.0 ** -16))
    assert np.array_equal(img[1], np.tile(np.array([7 / 64, 7 / 64, 7 / 64, 1.0], np.float32), (w, 1)))
    g = tmp_path / "narrow.hdr"                             # fewer than eight columns: always flat
    g.write_bytes(b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 4\n" + flat[:16])
    assert np.array_equal(load_hdr(str(g))[0], img[0, :4])
    a, b = load_hdr(os.path.join(GOLDEN, "sky.hdr")), load_hdr(os.path.join(GOLDEN, "sky_flat.hdr"))
    assert a.shape == (48, 96, 4) and np.array_equal(a, b) and a[..., :3].max() > 3000 and (a[43, :, :3] == 0).all()
    for bad in (b"P6\n", head[:20], head + rle[:10]):
        (tmp_path / "bad.hdr").write_bytes(bad)
        with pytest.raises((ValueError, IndexError)):
            load_hdr(str(tmp_path / "bad.hdr"))
