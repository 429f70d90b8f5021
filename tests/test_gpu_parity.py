"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs,
against the reference's golden images, and - at the benchmark's full size - through size-independent
properties (sharded == unsharded, determinism, accumulation linearity)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

FEATURES = {"distance": 5, "world-pos": 3, "view-pos": 4, "world-normal": 1, "view-normal": 2, "albedo": 0}
# fp32 tolerance for radiance: HIP and oracle evaluate the same expression trees, so differences come only from
# libm (sin/cos/pow/atan2 differ by ulps) and are amplified by path divergence in a few pixels.
REL_TOL = 1e-2          # per-pixel relative tolerance (with +1e-2 absolute floor)
MAX_BAD_FRACTION = 2e-3  # pixels allowed outside REL_TOL (paths that flipped a discrete decision)


@pytest.fixture(scope="module")
def R():
    from tauray_amd import renderer
    return renderer


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context(0)
    yield c


def _dup(size):
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    return DistributionParams(tuple(size), DISTRIBUTION_DUPLICATE, 0, 1, True)


def _render_hip(R, ctx, ss, scene, size, frames=1, viewports=1, dist=None, ieee=None, specialize=None, **kw):
    """`ieee`: True = the shading kernels at IEEE fp32 (trhip_pt_set_shading_arithmetic), None = the stage's default (the accuracy
    Vulkan asks of the reference's GLSL); `specialize`: False = the general kernels, None = the stage's default (a program compiled
    for the option set, from the kernel cache __graft_entry__.build() filled or through hipRTC)."""
    from tauray_amd.distribution import get_distribution_target_size
    d = dist or _dup(size)
    opt = R.options_for_scene(scene, **kw)
    pt = R.PathTracerStage(ctx, ss, opt, d)
    if ieee is not None:
        pt.set_shading_arithmetic(ieee)
    if specialize is not None:
        pt.set_specialization(specialize)
    tw, th = get_distribution_target_size(d)
    color = ctx.alloc(viewports * tw * th * 16).zero()
    for _ in range(frames):
        pt.run(color, viewports)
    img = color.download((viewports, th, tw, 4))
    c = pt.counters()
    assert c["stack_overflows"] == 0
    pt.close()
    return img


def _compare(img, ref, what, max_bad=MAX_BAD_FRACTION, strict=False):
    """`strict` (frames rendered with IEEE shading arithmetic, which follows the oracle expression by expression): the image mean is
    taken over the whole image.  Otherwise (the default arithmetic) it is taken over the pixels inside the tolerance, capped - and
    what the excluded pixels hold is bounded on its own."""
    assert np.isfinite(img).all(), f"{what}: non-finite output"
    err = np.abs(img[..., :3] - ref[..., :3])
    rel = err / (np.abs(ref[..., :3]) + 1e-2)
    off = rel.max(-1) > REL_TOL
    bad = float(off.mean())
    assert bad <= max_bad, f"{what}: {bad:.4%} pixels differ by more than {REL_TOL}"
    total = max(float(np.abs(ref[..., :3]).sum()), 1e-6)
    if strict:
        # the image mean catches a small bias everywhere, which the per-pixel tolerance would let through
        mean_err = abs(float(img[..., :3].mean()) - float(ref[..., :3].mean())) / max(float(ref[..., :3].mean()), 1e-6)
        assert mean_err < 2e-3, f"{what}: mean radiance off by {mean_err:.3e}"
    else:
        # The mean is taken over the pixels inside the tolerance, with values capped at a hundred times the frame's mean: one pixel
        # can hold a highlight thousands of times the mean (a sphere light in a near-mirror panel: GGX's D at small roughness
        # amplifies an ulp of n.h to 1-2 %, 7193 vs 7338 or 15632 vs 15772 in frames of mean 2.4-3.0 - found by
        # tools/fuzz_campaign.sh, profiles/r3/fuzz_campaign.txt; bit-equal to the oracle under IEEE shading arithmetic) and would
        # decide the statistic alone ...
        good = ~off
        cap = 100.0 * max(float(np.abs(ref[..., :3]).mean()), 1e-6)
        a, b = np.minimum(img[..., :3][good], cap), np.minimum(ref[..., :3][good], cap)
        mean_err = abs(float(a.mean()) - float(b.mean())) / max(float(b.mean()), 1e-6)
        assert mean_err < 2e-3, f"{what}: mean radiance off by {mean_err:.3e}"
        # ... and the error the excluded pixels carry is bounded against the light in the frame (those highlight pixels: 2e-3), with
        # the same cap on both images: in an option set whose estimator has fireflies (hemisphere sampling towards sphere lights that
        # next event estimation cannot reach: one path in a 96 x 96 frame carrying 4.5 % of the frame's light) a path that flips is
        # its firefly, whatever the size - two draws of 3 200, tools/fuzz_campaign.sh seeds 9101 / 9102, identical under the round-4
        # library (profiles/r5/fuzz_campaign.txt).  Capped, a flipped pixel can still cost 100 / pixels of the frame's light.
        capped_err = np.abs(np.minimum(img[..., :3], cap) - np.minimum(ref[..., :3], cap))
        excluded = float(capped_err[off].sum()) / max(float(np.minimum(np.abs(ref[..., :3]), cap).sum()), 1e-6)
        assert excluded < 2e-2, (f"{what}: the pixels outside the tolerance differ by {excluded:.3e} of the frame's light "
                                 f"(uncapped: {float(err[off].sum()) / total:.3e})")
    assert np.array_equal(img[..., 3], ref[..., 3]), f"{what}: alpha differs"


def _compare_both(R, ctx, ss, scene, size, ref, what, max_bad=MAX_BAD_FRACTION, max_bad_default=None, **kw):
    """The HIP path against the oracle's frame in both shading arithmetics: IEEE fp32 (strict comparison) and the default."""
    _compare(_render_hip(R, ctx, ss, scene, size, ieee=True, **kw), ref, what + " [IEEE shading]", max_bad=max_bad, strict=True)
    _compare(_render_hip(R, ctx, ss, scene, size, **kw), ref, what + " [default shading arithmetic]", max_bad=max_bad_default or max_bad)


@pytest.fixture(scope="module")
def glb128(R, ctx, test_glb_128):
    # A device holds one scene: the tests that take this fixture come first in the file, before any test uploads a scene of its own
    # to `ctx` (every later test builds its SceneStage itself).
    return R.SceneStage(ctx, test_glb_128)


# ----------------------------------------------------------------------------------------------------------
def test_extension_is_loaded_in_tree():
    from tauray_amd import _lib
    assert os.path.dirname(_lib.LIB_PATH) == os.path.dirname(_lib.__file__)
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "libtrhip.so" in maps


def test_tri_lights_bit_exact(glb128, oracle_scene_128):
    g, o = glb128.tri_lights(), oracle_scene_128.tri_lights()
    assert len(g) == 4096 and np.array_equal(g.view(np.uint8), o.view(np.uint8))


@pytest.mark.parametrize("name", list(FEATURES))
def test_feature_images(R, ctx, test_glb_512, oracle_scene_512, name):
    ss = R.SceneStage(ctx, test_glb_512)
    fs = R.FeatureStage(ctx, ss, FEATURES[name], _dup((512, 512)))
    buf = ctx.alloc(512 * 512 * 16).zero()
    fs.run(buf)
    img = buf.download((512, 512, 4))
    ref = oracle_scene_512.render_feature(FEATURES[name], 512, 512)
    gold = load_golden(name)
    tol = np.abs(gold) * 2.0 ** -10 + 2e-3
    assert (np.abs(img[..., :3] - gold) > tol).any(-1).sum() == 0, "outside the reference golden's half-float quantisation"
    if name == "albedo":   # sRGB decode uses powf: ulp-level differences
        assert np.abs(img - ref).max() < 1e-6
    else:                   # geometry is bit-exact against the oracle
        assert np.array_equal(img, ref)


def test_closest_hit_queries_bit_exact(R, ctx, glb128, oracle_scene_128):
    rng = np.random.default_rng(7)
    n = 200_000
    org = rng.uniform(-1.9, 1.9, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d, np.full((n, 1), np.inf, np.float32)], axis=1)
    rays[:100, 7] = rng.uniform(0.1, 2.0, size=100)          # finite tmax
    seeds = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    for sd, lights in ((seeds, True), (None, False)):
        g = glb128.trace_closest(rays, sd, include_lights=lights)
        o = oracle_scene_128.trace_closest(rays, sd, include_lights=lights)
        assert np.array_equal(g["instance_id"], o["instance_id"]) and np.array_equal(g["primitive_id"], o["primitive_id"])
        assert np.array_equal(g["t"].view(np.uint32), o["t"].view(np.uint32))
        hit = g["instance_id"] >= 0
        assert np.array_equal(g["bary_u"][hit].view(np.uint32), o["bary_u"][hit].view(np.uint32))
        assert np.array_equal(g["bary_v"][hit].view(np.uint32), o["bary_v"][hit].view(np.uint32))
        assert 0.5 < hit.mean() <= 1.0   # the room is open towards the camera


def test_axis_aligned_rays_on_box_planes(R, ctx, glb128, test_glb_128, oracle_scene_128):
    """Rays that do not move along one or two axes (+0 and -0 components), starting exactly on coordinates the geometry
    uses: the slab test then multiplies 0 by infinity on those axes and picks near / far planes by the sign of a zero.
    Hits must still equal the oracle's, whose traversal orders the planes with min / max instead."""
    from tauray_amd.scene import from_glm
    rng = np.random.default_rng(11)
    n = 60_000
    verts = test_glb_128.vertices
    sp = test_glb_128.spans
    pos = []
    for i in range(len(sp)):       # world-space vertex positions
        v = verts["pos"][sp["vertex_offset"][i]:sp["vertex_offset"][i] + sp["vertex_count"][i]]
        m = from_glm(test_glb_128.instances["model"][i])
        pos.append((np.c_[v, np.ones(len(v), np.float32)] @ np.asarray(m).T)[:, :3])
    pos = np.concatenate(pos).astype(np.float32)
    org = pos[rng.integers(0, len(pos), size=n)].copy()
    keep = rng.integers(0, 3, size=n)                       # one coordinate stays exactly on a vertex coordinate
    jitter = rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32)
    jitter[np.arange(n), keep] = 0
    org += jitter
    d = rng.normal(size=(n, 3)).astype(np.float32)
    zero = rng.integers(0, 3, size=n)
    d[np.arange(n), zero] = np.where(rng.uniform(size=n) < 0.5, np.float32(0.0), np.float32(-0.0))
    two = rng.uniform(size=n) < 0.3                         # a third of the rays move along a single axis
    zero2 = (zero + 1 + rng.integers(0, 2, size=n)) % 3
    d[np.arange(n)[two], zero2[two]] = np.float32(-0.0)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d, np.full((n, 1), np.inf, np.float32)], axis=1).astype(np.float32)
    g = glb128.trace_closest(rays, None, include_lights=False)
    o = oracle_scene_128.trace_closest(rays, None, include_lights=False)
    assert np.array_equal(g["instance_id"], o["instance_id"]) and np.array_equal(g["primitive_id"], o["primitive_id"])
    assert np.array_equal(g["t"].view(np.uint32), o["t"].view(np.uint32))
    assert (g["instance_id"] >= 0).mean() > 0.3
    srays = rays.copy()
    srays[:, 7] = rng.uniform(0.05, 3.0, size=n)
    gs, os_ = glb128.trace_shadow(srays), oracle_scene_128.trace_shadow(srays)
    assert np.array_equal(gs == 0, os_ == 0) and np.allclose(gs, os_, atol=1e-6)


def test_shadow_queries(R, ctx, glb128, oracle_scene_128):
    rng = np.random.default_rng(8)
    n = 100_000
    org = rng.uniform(-1.9, 1.9, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d, rng.uniform(0.05, 3.0, size=(n, 1)).astype(np.float32)], axis=1)
    g, o = glb128.trace_shadow(rays), oracle_scene_128.trace_shadow(rays)
    assert np.array_equal(g == 0, o == 0)
    assert np.allclose(g, o, atol=1e-6)           # partial coverage: product order may differ
    assert ((g > 0) & (g < 1)).sum() > 10          # the alpha-blended plane produces fractional visibility


def test_degenerate_rays_are_cheap_misses(R, ctx, glb128):
    rays = np.zeros((6, 8), dtype=np.float32)
    rays[:, 7] = np.inf
    rays[0, 4:7] = 0                                # zero direction (failed refraction sample, ggx.glsl:343-348)
    rays[1, 4:7] = np.nan
    rays[2, 4:7] = (np.inf, 0, 0)
    rays[3, 0:3] = np.nan; rays[3, 4:7] = (0, 0, -1)
    rays[4, 4:7] = (0, 0, -1)                       # a valid ray for contrast
    rays[5, 4:7] = (0, -0.0, 0)
    hits = glb128.trace_closest(rays, np.arange(6, dtype=np.uint32), include_lights=True)
    assert list(hits["instance_id"] >= 0) == [False, False, False, False, True, False]
    assert np.array_equal(glb128.trace_shadow(rays), np.array([1, 1, 1, 1, 0, 1], dtype=np.float32))


from tauray_amd.presets import NAMED_OPTION_SETS as OPTION_SETS      # noqa: E402  (the sets __graft_entry__.build() compiles programs for)


@pytest.mark.parametrize("name", list(OPTION_SETS))
def test_path_tracer_matches_oracle(R, ctx, glb128, test_glb_128, oracle, oracle_scene_128, name):
    kw = OPTION_SETS[name]
    scene = test_glb_128
    if name == "dof":
        scene.cameras[0].set_focus(0.2, 6.0, 6, 15.0, 0.036)
        glb128.update_cameras(scene.cameras)
        osc = oracle.OracleScene(scene)
    else:
        osc = oracle_scene_128
    try:
        ref = osc.render_pt(oracle.options_for_scene(scene, **kw), 128, 128)
        _compare_both(R, ctx, glb128, scene, (128, 128), ref, name, **kw)
        if name in ("sobol-owen", "blackman-harris", "tri-hybrid", "regularization+clamp", "no-nee", "dof"):
            # the program compiled for the option set against the general kernels, which take the set as data: the same bits at IEEE
            # fp32; at the default arithmetic every instance is an implementation of its own within Vulkan's accuracy (which of
            # 1 / sqrt(x) becomes v_rsq_f32 and which v_rcp_f32(v_sqrt_f32) depends on the code around it): equal to a few ulps per
            # operation, i.e. to the comparator's tolerance with nearly every pixel inside 1e-4
            a, b = (_render_hip(R, ctx, glb128, scene, (128, 128), ieee=True, specialize=sp, **kw) for sp in (None, False))
            assert np.array_equal(a, b), f"{name}: specialised != general at IEEE fp32"
            a, b = (_render_hip(R, ctx, glb128, scene, (128, 128), specialize=sp, **kw) for sp in (None, False))
            _compare(a, b, f"{name}: specialised vs general kernels, default arithmetic")
            assert (np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-6).all(-1).mean() > 0.98
    finally:
        if name == "dof":
            scene.cameras[0].focus = (1.0, 0.0, 0.0, 0.0)
            glb128.update_cameras(scene.cameras)


def test_progressive_accumulation_over_frames(R, ctx, glb128, test_glb_128, oracle, oracle_scene_128):
    """--accumulation: frame k mixes into the running mean with samples_accumulated (gbuffer.glsl:18-28)."""
    opt = R.options_for_scene(test_glb_128, max_bounces=3, samples_per_pixel=2)
    pt = R.PathTracerStage(ctx, glb128, opt, _dup((128, 128)))
    color = ctx.alloc(128 * 128 * 16).zero()
    ref = np.zeros((1, 128, 128, 4), dtype=np.float32)
    oopt = oracle.options_for_scene(test_glb_128, max_bounces=3, samples_per_pixel=2)
    for f in range(3):
        pt.run(color)
        ref = oracle_scene_128.render_pt(oopt, 128, 128, frame_counter=f, samples_accumulated=2 * f, color=ref)
    _compare(color.download((1, 128, 128, 4)), ref, "accumulated frames")
    # reset_accumulated_samples keeps the sample counter (offline frames, src/tauray.cc:1101)
    pt.reset_accumulated_samples()
    pt.run(color)
    ref2 = oracle_scene_128.render_pt(oopt, 128, 128, frame_counter=3, samples_accumulated=0)
    _compare(color.download((1, 128, 128, 4)), ref2, "frame after reset")


@pytest.mark.parametrize("name,kw,frames", [
    ("1-sample", dict(samples_per_pixel=1), 1),
    ("4-per-pass-sobol", dict(samples_per_pixel=8, samples_per_pass=4, sampler=1), 1),
    ("accumulated-area-lights", dict(samples_per_pixel=2, samples_per_pass=2, tri_light_mode=0, film=1, film_radius=0.5), 2),
])
def test_direct_stage_matches_oracle(R, ctx, glb128, test_glb_128, oracle, oracle_scene_128, name, kw, frames):
    """direct_stage (src/direct_stage.cc, shader/direct.rgen): first hit with sphere lights hidden + samples_per_pass light
    samples; colour, demodulated diffuse (divided by the sample count) and reflection (not divided, as the shader has it)
    and the first-hit AOVs against the oracle."""
    names = ["color", "diffuse", "reflection", "albedo", "normal", "pos", "instance_id"]
    opt = R.options_for_scene(test_glb_128, max_bounces=4, **kw)
    st = R.DirectStage(ctx, glb128, opt, _dup((128, 128)))
    bufs = {n: ctx.alloc(128 * 128 * R.PathTracerStage.TARGETS[n][0] * 4).zero() for n in names}
    for _ in range(frames):
        st.run_targets(bufs)
    got = {n: np.frombuffer(bufs[n].download((1, 128, 128, R.PathTracerStage.TARGETS[n][0])).tobytes(), dtype=R.PathTracerStage.TARGETS[n][1])
              .reshape(1, 128, 128, R.PathTracerStage.TARGETS[n][0]) for n in names}
    c = st.counters()
    assert c["stack_overflows"] == 0 and c["closest_rays"] > 0 and c["shadow_rays"] > 0
    st.close()
    oopt = oracle.options_for_scene(test_glb_128, max_bounces=4, **kw)
    ref = None
    spp = kw["samples_per_pixel"]
    for f in range(frames):
        ref = oracle_scene_128.render_pt_targets(oopt, 128, 128, names, frame_counter=f, samples_accumulated=spp * f, targets=ref, direct=True)
    _compare(got["color"], ref["color"], f"direct {name}: color")
    assert np.array_equal(got["instance_id"], ref["instance_id"])
    for n, tol in (("albedo", 1e-6), ("normal", 1e-5), ("pos", 1e-5)):
        assert float(np.abs(got[n] - ref[n]).max()) <= tol * max(1.0, float(np.abs(ref[n]).max())), f"direct {name}: {n}"
    for n in ("diffuse", "reflection"):
        rel = np.abs(got[n][..., :3] - ref[n][..., :3]) / (np.abs(ref[n][..., :3]) + 1e-2)
        assert float((rel.max(-1) > REL_TOL).mean()) <= MAX_BAD_FRACTION, f"direct {name}: {n}"
        assert float((np.abs(got[n][..., 3] - ref[n][..., 3]) > 1e-4 * (np.abs(ref[n][..., 3]) + 1.0)).mean()) <= MAX_BAD_FRACTION, f"direct {name}: {n} alpha"
    # sanity: direct light only - darker than the path tracer's image, and not black
    assert 0.01 < float(got["color"][..., :3].mean())


def test_pre_transformed_vertices(R, ctx, glb128, test_glb_128, oracle, oracle_scene_128):
    """PRE_TRANSFORMED_VERTICES (--pre-transform-vertices): shading reads scene_stage's world-space vertex copy
    (shader/pre_transform.comp:26-42) and skips the model / normal-matrix transforms (shader/rt.glsl:18-22)."""
    kw = dict(max_bounces=4, pre_transformed_vertices=1)
    names = ["color", "normal", "pos", "instance_id"]
    got = _render_targets_hip(R, ctx, glb128, test_glb_128, (128, 128), names, **kw)
    ref = oracle_scene_128.render_pt_targets(oracle.options_for_scene(test_glb_128, **kw), 128, 128, names)
    _compare(got["color"], ref["color"], "pre-transformed vertices")
    assert np.array_equal(got["instance_id"], ref["instance_id"])
    assert float(np.abs(got["pos"] - ref["pos"]).max()) <= 1e-5 * float(np.abs(ref["pos"]).max())
    assert float(np.abs(got["normal"] - ref["normal"]).max()) <= 1e-5
    # against the default path the image is the same up to the rounding of interpolate-then-transform vs transform-then-interpolate
    base = _render_targets_hip(R, ctx, glb128, test_glb_128, (128, 128), ["color", "pos"], max_bounces=4)
    assert float(np.abs(got["pos"] - base["pos"]).max()) < 1e-4
    assert abs(float(got["color"][..., :3].mean()) - float(base["color"][..., :3].mean())) < 0.02 * float(base["color"][..., :3].mean())


def _render_targets_hip(R, ctx, ss, scene, size, names, frames=1, **kw):
    opt = R.options_for_scene(scene, **kw)
    pt = R.PathTracerStage(ctx, ss, opt, _dup(size))
    w, h = size
    bufs = {n: ctx.alloc(w * h * R.PathTracerStage.TARGETS[n][0] * 4).zero() for n in names}
    for _ in range(frames):
        pt.run_targets(bufs)
    out = {}
    for n in names:
        ch, dt = R.PathTracerStage.TARGETS[n]
        out[n] = np.frombuffer(bufs[n].download((1, h, w, ch)).tobytes(), dtype=dt).reshape(1, h, w, ch)
    assert pt.counters()["stack_overflows"] == 0
    pt.close()
    return out


@pytest.mark.parametrize("name,kw,frames", [
    ("default", dict(max_bounces=4), 1),
    ("2-per-pass-accumulated", dict(max_bounces=3, samples_per_pixel=4, samples_per_pass=2), 2),
    ("white-albedo-clamp", dict(max_bounces=4, use_white_albedo_on_first_bounce=1, indirect_clamping=4.0), 1),
])
def test_gbuffer_targets_match_oracle(R, ctx, glb128, test_glb_128, oracle, oracle_scene_128, name, kw, frames):
    """write_all_outputs (path_tracer.glsl:535-576): every gbuffer target the path tracer can write, against the oracle:
    demodulated diffuse / reflection (running means, a = 1/length of the second segment) and the first-hit AOVs."""
    names = list(R.PathTracerStage.TARGETS)
    got = _render_targets_hip(R, ctx, glb128, test_glb_128, (128, 128), names, frames=frames, **kw)
    oopt = oracle.options_for_scene(test_glb_128, **kw)
    ref = None
    spp = kw.get("samples_per_pixel", 1)
    for f in range(frames):
        ref = oracle_scene_128.render_pt_targets(oopt, 128, 128, names, frame_counter=f, samples_accumulated=spp * f, targets=ref)
    _compare(got["color"], ref["color"], f"{name}: color")
    # first-hit AOVs: same arithmetic on bit-equal hits; textures / pow differ by ulps at most
    assert np.array_equal(got["instance_id"], ref["instance_id"]), "instance id target"
    for n, tol in (("albedo", 1e-6), ("material", 1e-6), ("normal", 1e-5), ("pos", 1e-5)):
        d = np.abs(got[n] - ref[n])
        assert float(d.max()) <= tol * max(1.0, float(np.abs(ref[n]).max())), f"{name}: {n} target differs by {float(d.max()):.3e}"
    # demodulated light: same tolerance model as radiance; alpha (1/length) is exact arithmetic on the same hit points
    for n in ("diffuse", "reflection"):
        g, r = got[n], ref[n]
        assert np.isfinite(g).all()
        rel = np.abs(g[..., :3] - r[..., :3]) / (np.abs(r[..., :3]) + 1e-2)
        assert float((rel.max(-1) > REL_TOL).mean()) <= MAX_BAD_FRACTION, f"{name}: {n} rgb"
        bad_a = np.abs(g[..., 3] - r[..., 3]) > 1e-4 * (np.abs(r[..., 3]) + 1.0)
        assert float(bad_a.mean()) <= MAX_BAD_FRACTION, f"{name}: {n} alpha"
    # the colour target is emission + modulate_color(first hit, diffuse, reflection) (path_tracer.rgen:112, material.glsl:57-65)
    if frames == 1 and spp == 1:
        alb = np.ones_like(got["albedo"][..., :3]) if kw.get("use_white_albedo_on_first_bounce") else got["albedo"][..., :3]
        met = got["material"][..., 0:1]
        mod = got["diffuse"][..., :3] * alb * (1 - met) + got["reflection"][..., :3] * (0.02 * (1 - met) + alb * met) / (0.02 * (1 - met) + met)
        resid = got["color"][..., :3] - mod      # = first-hit emission: zero wherever the first hit neither emits nor misses
        lit = (got["instance_id"][..., 0] >= 0) & (np.abs(resid).max(-1) < 1e-3)
        assert lit.mean() > 0.5


@pytest.mark.parametrize("world,strategy", [(2, 1), (3, 1), (8, 1), (3, 2)])
def test_fake_device_sharding_is_bitwise_identical(R, ctx, glb128, test_glb_128, world, strategy):
    """The reference tests multi-GPU with --fake-devices (src/context.cc:415-416): render every device's share on
    one GPU, stitch, compare with the unsharded frame.  Pixel -> RNG mapping is by absolute pixel."""
    from tauray_amd import distribution as D
    W, H = 128, 128
    full = _render_hip(R, ctx, glb128, test_glb_128, (W, H), max_bounces=3)
    dists, cum = [], 0.0
    for i in range(world):
        dists.append(D.get_device_distribution_params((W, H), strategy, cum, 1.0 / world, i, world, i == 0))
        cum += 1.0 / world
    opt = R.options_for_scene(test_glb_128, max_bounces=3)
    primary = ctx.alloc(W * H * 16).zero()
    primary2 = ctx.alloc(W * H * 16).zero()
    stitch = R.StitchStage(ctx, (W, H))
    parts = []
    for i, d in enumerate(dists):
        pt = R.PathTracerStage(ctx, glb128, opt, d)
        if i == 0:
            pt.run(primary)
            pt.reset_accumulated_samples(); pt.reset_sample_counter()
            pt.run(primary2)
        else:
            tw, th = D.get_distribution_target_size(d)
            part = ctx.alloc(tw * th * 16).zero()
            pt.run(part)
            stitch.run_one(d, part, primary)
            parts.append(part)
        pt.close()
    got = primary.download((1, H, W, 4))
    assert np.array_equal(got, full), f"{(got != full).any(-1).sum()} pixels differ"
    stitch.run_all(dists[1:], parts, primary2)          # all partials in one launch (trhip_stitch_batch)
    assert np.array_equal(primary2.download((1, H, W, 4)), full)


def test_viewports_and_camera_grid(R, ctx, test_glb_128, oracle):
    """Viewport (array layer) batching, config 5's light-field grid: layer v == single render with camera v."""
    import copy
    from tauray_amd import scene as S
    scene = copy.copy(test_glb_128)
    scene.cameras = S.generate_camera_grid(test_glb_128.cameras[0], 3, 2, 0.3, 0.3, 5.0)
    ss = R.SceneStage(ctx, scene)
    img = _render_hip(R, ctx, ss, scene, (128, 128), viewports=6, max_bounces=2)
    osc = oracle.OracleScene(scene)
    ref = osc.render_pt(oracle.options_for_scene(scene, max_bounces=2), 128, 128, viewports=6)
    assert img.shape == (6, 128, 128, 4)
    _compare(img, ref, "camera grid")
    assert not np.array_equal(img[0], img[5])


def test_envmap_scene_matches_oracle(R, ctx, oracle):
    """Sponza-class scene: environment-map NEE (alias table), directional sun, emissive quads, alpha-tested curtains."""
    from tauray_amd import scenes
    scene = scenes.sponza_class(seed=3, target_tris=40000, width=160, height=90)
    rng = np.random.default_rng(0)
    scene.envmap = (rng.uniform(0.2, 1.0, size=(8, 16, 4)) ** 2).astype(np.float32)      # non-uniform sky
    scene.envmap[1, 3, :3] = 20.0
    ss = R.SceneStage(ctx, scene)
    osc = oracle.OracleScene(scene)
    for kw in (dict(max_bounces=4), dict(max_bounces=3, sampler=1)):
        img = _render_hip(R, ctx, ss, scene, (160, 90), **kw)
        ref = osc.render_pt(oracle.options_for_scene(scene, **kw), 160, 90)
        _compare(img, ref, f"sponza_class {kw}")
        assert img[..., :3].mean() > 0.01


def test_bench_scene_one_million_triangles(R, ctx, oracle):
    """BASELINE config 4's scene exactly as bench.py renders it (sponza_teapots: 1 M triangles, glass / metal / diffuse
    teapots, alpha-tested curtains, sun + emissive quads), at a quarter of the resolution so the oracle finishes in seconds:
    primary hits bit-equal on the 4-wide PLOC tree vs the oracle's SAH tree, radiance within the fp32 tolerance."""
    from tauray_amd import scenes
    W, H = 480, 272
    scene = scenes.sponza_teapots(width=W, height=H)
    assert scene.triangle_count > 900_000
    ss = R.SceneStage(ctx, scene)
    assert ss.accel["node_count"] == ss.accel["leaf_count"] - 1 and ss.accel["leaf_count"] == scene.triangle_count
    osc = oracle.OracleScene(scene)
    for fid in (5, 9):   # hit distance, (instance, primitive)
        fs = R.FeatureStage(ctx, ss, fid, _dup((W, H)))
        buf = ctx.alloc(W * H * 16).zero()
        fs.run(buf)
        g, r = buf.download((H, W, 4)), osc.render_feature(fid, W, H)
        diff = ~((g == r) | (np.isnan(g) & np.isnan(r)))     # misses keep the NaN default value
        assert not diff.any(), f"feature {fid} on the 1 M triangle scene: {int(diff.any(-1).sum())} pixels differ, max {np.nanmax(np.abs(g - r)):.3e}"
    # incoherent rays from inside the atrium (what bounces 1+ look like), stochastic alpha on: hits bit-equal
    rng = np.random.default_rng(21)
    n = 100_000
    lo, hi = np.array(ss.accel["bounds_min"], np.float32), np.array(ss.accel["bounds_max"], np.float32)
    org = (lo + (hi - lo) * rng.uniform(0.05, 0.95, size=(n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d, np.full((n, 1), np.inf, np.float32)], axis=1)
    seeds = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    g, o = ss.trace_closest(rays, seeds), osc.trace_closest(rays, seeds)
    assert np.array_equal(g["instance_id"], o["instance_id"]) and np.array_equal(g["primitive_id"], o["primitive_id"])
    assert np.array_equal(g["t"].view(np.uint32), o["t"].view(np.uint32)) and (g["instance_id"] >= 0).mean() > 0.9
    srays = rays.copy()
    srays[:, 7] = rng.uniform(0.1, 20.0, size=n)
    gs, os_ = ss.trace_shadow(srays), osc.trace_shadow(srays)
    assert np.array_equal(gs == 0, os_ == 0) and np.allclose(gs, os_, atol=1e-6)
    kw = dict(max_bounces=4)
    img = _render_hip(R, ctx, ss, scene, (W, H), **kw)
    ref = osc.render_pt(oracle.options_for_scene(scene, **kw), W, H)
    _compare(img, ref, "sponza_teapots 4 bounces")


def test_other_projections(R, ctx, test_glb_128, oracle):
    import copy
    from tauray_amd import scene as S
    for proj in (S.PROJ_ORTHOGRAPHIC, S.PROJ_EQUIRECTANGULAR):
        scene = copy.copy(test_glb_128)
        cam = copy.deepcopy(test_glb_128.cameras[0])
        cam.projection = proj
        if proj == S.PROJ_ORTHOGRAPHIC:
            cam.ortho = (-2.0, 2.0, -2.0, 2.0, 0.0, 100.0)
        else:
            cam.transform = np.eye(4)
        scene.cameras = [cam]
        ss = R.SceneStage(ctx, scene)
        img = _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=2, projection=proj)
        ref = oracle.OracleScene(scene).render_pt(oracle.options_for_scene(scene, max_bounces=2, projection=proj), 128, 128)
        _compare(img, ref, f"projection {proj}")


def test_edge_scenes(R, ctx, oracle):
    """Empty scene, single triangle, degenerate triangles."""
    from tauray_amd import scene as S
    cam = S.Camera(fov=60, aspect=1.0)
    cam.transform = S.trs_matrix((0, 0, 3))
    tri = np.zeros(3, dtype=S.VERTEX)
    tri["pos"] = [(-1, -1, 0), (1, -1, 0), (0, 1, 0)]
    tri["normal"] = (0, 0, 1)
    tri["tangent"] = (1, 0, 0, 1)
    mat = S.make_material(albedo=(0.8, 0.8, 0.8, 1), metallic=0.0, roughness=0.5, emission=(0.5, 0.2, 0.1))
    light = S.make_point_light((5, 5, 5), (0, 0, 2), 0.2)

    def scene_with(verts, idx):
        n = len(idx) // 3
        inst = S.make_instance(np.eye(4), mat) if n else np.zeros(0, dtype=S.INSTANCE)
        return S.SceneDesc(instances=inst, spans=np.array([(0, len(verts), 0, n)] if n else [], dtype=S.MESH_SPAN), vertices=verts,
                           indices=np.asarray(idx, dtype=np.uint32), point_lights=light, cameras=[cam]).finalize(True)

    deg = np.zeros(6, dtype=S.VERTEX)
    deg[:3] = tri
    deg["pos"][3:] = (0.25, 0.25, 0.5)       # zero-area triangle in front of the real one
    deg["normal"][3:] = (0, 0, 1)
    for what, sc in (("empty", scene_with(np.zeros(0, dtype=S.VERTEX), [])), ("single triangle", scene_with(tri, [0, 1, 2])),
                     ("degenerate", scene_with(deg, [0, 1, 2, 3, 4, 5]))):
        ss = R.SceneStage(ctx, sc)
        img = _render_hip(R, ctx, ss, sc, (64, 64), max_bounces=3)
        ref = oracle.OracleScene(sc).render_pt(oracle.options_for_scene(sc, max_bounces=3), 64, 64)
        _compare(img, ref, what)
        # the in-place update of the acceleration structure copes with the same corner cases (no nodes, one leaf)
        ss.update_instances(sc.instances, refit=True)
        assert np.array_equal(_render_hip(R, ctx, ss, sc, (64, 64), max_bounces=3), img), f"{what}: refit changed the frame"
    assert ss.accel["triangle_count"] == 2

    # coincident surfaces: the same two-triangle quad three times (two instances at the same place with different materials,
    # the third with its triangles listed twice).  Equal hit distances go to the lower (instance, primitive) in both
    # implementations, whatever order the trees present the candidates in.
    quad = np.zeros(4, dtype=S.VERTEX)
    quad["pos"] = [(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)]
    quad["normal"] = (0, 0, 1)
    quad["tangent"] = (1, 0, 0, 1)
    mats = [S.make_material(albedo=c + (1,), metallic=0.0, roughness=0.6) for c in ((0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9))]
    insts = np.concatenate([S.make_instance(np.eye(4), m) for m in mats])
    verts = np.concatenate([quad, quad, quad])
    idx = np.array([0, 1, 2, 0, 2, 3] + [0, 1, 2, 0, 2, 3] + [0, 2, 3, 0, 1, 2, 0, 1, 2, 0, 2, 3], dtype=np.uint32)
    spans = np.array([(0, 4, 0, 2), (4, 4, 6, 2), (8, 4, 12, 4)], dtype=S.MESH_SPAN)
    sc = S.SceneDesc(instances=insts, spans=spans, vertices=verts, indices=idx, point_lights=light, cameras=[cam]).finalize(True)
    ss = R.SceneStage(ctx, sc)
    osc = oracle.OracleScene(sc)
    for fid in (9, 5, 0):       # instance id, distance, albedo
        fs = R.FeatureStage(ctx, ss, fid, _dup((64, 64)))
        buf = ctx.alloc(64 * 64 * 16).zero()
        fs.run(buf)
        assert np.array_equal(buf.download((64, 64, 4)), osc.render_feature(fid, 64, 64), equal_nan=True), f"coincident quads, feature {fid}"
        if fid == 9:
            ids = buf.download((64, 64, 4))[..., 0]
            assert set(np.unique(ids[np.isfinite(ids)])) == {0.0}, "the lowest instance wins every tie"
    _compare(_render_hip(R, ctx, ss, sc, (64, 64), max_bounces=3), osc.render_pt(oracle.options_for_scene(sc, max_bounces=3), 64, 64), "coincident quads")


def test_tonemap_operators(R, ctx, oracle):
    rng = np.random.default_rng(4)
    x = (rng.uniform(0, 1, size=(2, 33, 47, 4)) ** 3 * 20).astype(np.float32)
    x[..., 3] = rng.uniform(0, 1, size=x.shape[:-1])
    src = ctx.alloc(x.nbytes).upload(x)
    dst = ctx.alloc(x.nbytes)
    for op in range(5):
        R.TonemapStage(ctx, op=op, exposure=1.3, gamma=2.2).run(src, dst, 47, 33, 2)
        got = dst.download(x.shape)
        ref = oracle.tonemap(x, op=op, exposure=1.3, gamma=2.2)
        assert np.allclose(got, ref, rtol=2e-6, atol=1e-6), f"operator {op}"
    R.TonemapStage(ctx, op=2, alpha_grid_background=True).run(src, dst, 47, 33, 2)
    assert np.isfinite(dst.download(x.shape)).all()
    # values a renderer can hand over: zero, negative, huge, infinite and NaN radiance (the operators clamp to [0, 1000]
    # before their curve, shader/tonemap.glsl:35-55; what a NaN becomes is whatever min / max make of it in both)
    y = x.copy()
    y[0, 0, :8, :3] = [[0, 0, 0], [-1, -0.5, -1e-3], [1e30, 1e20, 5e4], [np.inf, 1, 0], [np.nan, 0.5, 0.25], [1000, 1000, 1000], [1e-30, 1e-38, 1e-45], [0.004, 0.0039, 0.0041]]
    src.upload(y)
    for op in range(5):
        R.TonemapStage(ctx, op=op, exposure=1.0, gamma=2.2).run(src, dst, 47, 33, 2)
        got, ref = dst.download(y.shape), oracle.tonemap(y, op=op, exposure=1.0, gamma=2.2)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f"operator {op}: NaNs in different places"
        assert np.allclose(got, ref, rtol=2e-6, atol=1e-6, equal_nan=True), f"operator {op}, special values"


def test_api_errors(R, ctx, test_glb_128):
    """Errors surface as exceptions with the library's message (the reference throws std::runtime_error)."""
    c2 = R.Context(0)
    ss = R.SceneStage(c2)            # nothing uploaded
    pt = R.PathTracerStage(c2, ss, R.make_options(), _dup((8, 8)))
    with pytest.raises(R.TrhipError, match="build_accel"):
        pt.run(c2.alloc(8 * 8 * 16))
    with pytest.raises(R.TrhipError, match="max_bounces"):
        R.PathTracerStage(c2, ss, R.make_options(max_bounces=0))
    with pytest.raises(R.TrhipError):
        R.Context(4096)
    import copy
    bad = copy.copy(test_glb_128)
    bad.indices = test_glb_128.indices.copy()
    bad.indices[5] = 10**6
    with pytest.raises(R.TrhipError, match="out of range"):
        R.SceneStage(c2, bad)
    bad = copy.copy(test_glb_128)
    bad.vertices = test_glb_128.vertices.copy()
    bad.vertices["pos"][100, 1] = np.nan
    with pytest.raises(R.TrhipError, match="non-finite vertex"):
        R.SceneStage(c2, bad)
    bad = copy.copy(test_glb_128)
    bad.instances = test_glb_128.instances.copy()
    bad.instances["model"][2][1, 3] = np.inf
    with pytest.raises(R.TrhipError, match="non-finite instance"):
        R.SceneStage(c2, bad)


# ---------------------------------------------------------------- full-size properties (BASELINE configs 2 and 4)
def test_full_size_properties(R, ctx):
    from tauray_amd import scenes
    from tauray_amd import distribution as D
    W, H = 1920, 1080
    scene = scenes.test_glb(W, H)
    ss = R.SceneStage(ctx, scene)
    a = _render_hip(R, ctx, ss, scene, (W, H), max_bounces=4)
    b = _render_hip(R, ctx, ss, scene, (W, H), max_bounces=4)
    assert np.isfinite(a).all() and np.array_equal(a, b), "a frame is a pure function of (scene, options, frame index)"
    assert (a[..., 3] == 1).all() and a[..., :3].min() >= 0
    # 8-way scanline sharding + stitch == unsharded (config 4's partitioning)
    opt = R.options_for_scene(scene, max_bounces=4)
    primary = ctx.alloc(W * H * 16).zero()
    stitch = R.StitchStage(ctx, (W, H))
    for i in range(8):
        d = D.get_device_distribution_params((W, H), D.DISTRIBUTION_SCANLINE, i / 8, 1 / 8, i, 8, i == 0)
        pt = R.PathTracerStage(ctx, ss, opt, d)
        if i == 0:
            pt.run(primary)
        else:
            tw, th = D.get_distribution_target_size(d)
            assert (tw, th) == (1920, 135)
            part = ctx.alloc(tw * th * 16)
            pt.run(part)
            stitch.run_one(d, part, primary)
        pt.close()
    assert np.array_equal(primary.download((1, H, W, 4)), a)
    # accumulation linearity: two accumulated 1-spp frames == mean of the two independent frames (up to 1 ulp of mix())
    pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
    acc = ctx.alloc(W * H * 16).zero()
    pt.run(acc); f0 = acc.download((1, H, W, 4)).copy()
    pt.run(acc); both = acc.download((1, H, W, 4))
    pt.reset_accumulated_samples(); pt.reset_sample_counter()
    pt.run(acc); pt.reset_accumulated_samples(); pt.run(acc); f1 = acc.download((1, H, W, 4))
    assert np.array_equal(f0, a)
    assert np.allclose(both[..., :3], 0.5 * (f0[..., :3] + f1[..., :3]), rtol=1e-6, atol=1e-7)
    # four accumulated 1-spp frames are the four passes of one 4-spp frame: same sample indices (sample_counter =
    # frame * spp, src/rt_stage.cc:81), same running mean (gbuffer.glsl:18-28)
    pt.reset_accumulated_samples(); pt.reset_sample_counter()
    for _ in range(4):
        pt.run(acc)
    four_frames = acc.download((1, H, W, 4))
    pt.close()
    one_frame = _render_hip(R, ctx, ss, scene, (W, H), max_bounces=4, samples_per_pixel=4, samples_per_pass=1)
    assert np.array_equal(four_frames, one_frame)


def test_dynamic_scene_rebuild_and_motion(R, ctx, oracle):
    """Dynamic scenes: new instance transforms -> trhip_scene_update_instances + rebuild; previous-frame transforms and
    cameras feed the motion features (src/feature_stage.cc:54-62) and the path tracer's screen-motion target
    (path_tracer.glsl:557-562).  Checked against a freshly built oracle scene with the same instance records."""
    import copy
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene import to_glm, from_glm
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), 128, 128)
    ss = R.SceneStage(ctx, scene)
    first = _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=2)
    _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=2, pre_transformed_vertices=1)    # builds the world-space vertex copy

    def feature(fid, sstage):
        fs = R.FeatureStage(ctx, sstage, fid, _dup((128, 128)))
        buf = ctx.alloc(128 * 128 * 16).zero()
        fs.run(buf)
        return buf.download((128, 128, 4))

    # frame 1: the teapot (instance 4) and Suzanne (5) move; the camera moved too
    for inst, (dx, ang) in ((4, (0.35, 0.4)), (5, (-0.2, -0.25))):
        old = from_glm(scene.instances["model"][inst])
        c, s_ = np.cos(ang), np.sin(ang)
        move = np.array([[c, 0, s_, dx], [0, 1, 0, 0.1], [-s_, 0, c, 0], [0, 0, 0, 1.0]])
        new = move @ old
        scene.instances["model_prev"][inst] = to_glm(old)
        scene.instances["model"][inst] = to_glm(new)
        scene.instances["model_normal"][inst] = to_glm(np.linalg.inv(new).T)
    prev_cams = copy.deepcopy(scene.cameras)
    prev_cams[0].transform = np.asarray(prev_cams[0].transform) @ np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1], [0, 0, 0, 1.0]])
    info = ss.update_instances(scene.instances)
    assert info["node_count"] > 0
    ss.set_previous_cameras(prev_cams)
    osc = oracle.OracleScene(scene)
    osc.set_previous_cameras(prev_cams)

    moved = _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=2)
    assert float(np.abs(moved - first).max()) > 0.05, "the rebuilt structure still shows the old frame"
    _compare(moved, osc.render_pt(oracle.options_for_scene(scene, max_bounces=2), 128, 128), "after instance update")
    for fid in (5, 3, 9):      # distance, world pos, instance id: bit-equal on the rebuilt structure
        assert np.array_equal(feature(fid, ss), osc.render_feature(fid, 128, 128)), f"feature {fid} after rebuild"
    for fid, tol in ((6, 1e-5), (7, 1e-5), (8, 1e-5)):   # world / view / screen motion
        g, r = feature(fid, ss), osc.render_feature(fid, 128, 128)
        hit = np.isfinite(r[..., 0])
        assert np.array_equal(hit, np.isfinite(g[..., 0]))
        assert float(np.abs(g[hit] - r[hit]).max()) <= tol * max(1.0, float(np.abs(r[hit]).max())), f"motion feature {fid}"
        if fid == 6:
            ids = feature(9, ss)[..., 0]
            assert float(np.abs(g[ids == 4][:, :3]).max()) > 0.1 and float(np.abs(g[ids == 0][:, :3]).max()) == 0.0
    got = _render_targets_hip(R, ctx, ss, scene, (128, 128), ["color", "screen_motion", "instance_id"], max_bounces=2)
    ref = osc.render_pt_targets(oracle.options_for_scene(scene, max_bounces=2), 128, 128, ["color", "screen_motion", "instance_id"])
    assert np.array_equal(got["instance_id"], ref["instance_id"])
    assert float(np.abs(got["screen_motion"] - ref["screen_motion"]).max()) <= 1e-5
    # frame 2: everything moves again, this time the tree is kept and refitted (an acceleration-structure update)
    for inst, (dx, ang) in ((4, (-0.5, -0.7)), (5, (0.3, 0.5)), (6, (0.0, 1.0))):
        old = from_glm(scene.instances["model"][inst])
        c, s_ = np.cos(ang), np.sin(ang)
        move = np.array([[c, -s_, 0, dx], [s_, c, 0, -0.05], [0, 0, 1, 0.15], [0, 0, 0, 1.0]])
        new = move @ old
        scene.instances["model_prev"][inst] = to_glm(old)
        scene.instances["model"][inst] = to_glm(new)
        scene.instances["model_normal"][inst] = to_glm(np.linalg.inv(new).T)
    info = ss.update_instances(scene.instances, refit=True)
    assert info["build_ms"] > 0
    osc2 = oracle.OracleScene(scene)
    for fid in (5, 3, 9):
        assert np.array_equal(feature(fid, ss), osc2.render_feature(fid, 128, 128)), f"feature {fid} after refit"
    assert np.array_equal(ss.tri_lights().view(np.uint8), osc2.tri_lights().view(np.uint8)), "tri lights after refit"
    refit_img = _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=3)
    _compare(refit_img, osc2.render_pt(oracle.options_for_scene(scene, max_bounces=3), 128, 128), "after refit")
    # the pre-transformed vertex copy is rebuilt for the new transforms (it was made for the old ones by the first frames)
    pre_before = _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=2, pre_transformed_vertices=1)
    _compare(pre_before, osc2.render_pt(oracle.options_for_scene(scene, max_bounces=2, pre_transformed_vertices=1), 128, 128), "pre-transformed after refit")
    ss.update_instances(scene.instances)          # a rebuild gives the same frame, bit for bit
    assert np.array_equal(refit_img, _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=3)), "refit vs rebuild"
    # the API refuses a different instance count, and rendering before the rebuild
    with pytest.raises(R.TrhipError):
        R._lib.check(R._lib.lib().trhip_scene_update_instances(ctx.h, scene.instances.ctypes.data, len(scene.instances) - 1))


def _bend_rig(vertices, phase):
    """A three-joint chain along +y for a mesh: SKIN records from the bind pose, joint matrices for one pose."""
    from tauray_amd.scene import SKIN
    y = vertices["pos"][:, 1].astype(np.float64)
    lo, hi = float(y.min()), float(y.max())
    t = (y - lo) / max(hi - lo, 1e-6) * 2.0          # 0..2 over the chain
    skins = np.zeros(len(vertices), dtype=SKIN)
    j0 = np.clip(np.floor(t), 0, 1).astype(np.uint32)
    f = np.clip(t - j0, 0.0, 1.0)
    f = f * f * (3 - 2 * f)
    skins["joints"][:, 0] = j0; skins["joints"][:, 1] = j0 + 1; skins["joints"][:, 2] = 2; skins["joints"][:, 3] = 0
    skins["weights"][:, 0] = 1 - f; skins["weights"][:, 1] = f

    def about(pivot_y, ang, shift=(0, 0, 0), scale=1.0):
        c, s = np.cos(ang), np.sin(ang)
        r = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]) @ np.diag([scale, 1.0, scale, 1.0])
        to, back = np.eye(4), np.eye(4)
        to[1, 3], back[1, 3] = -pivot_y, pivot_y
        mv = np.eye(4); mv[:3, 3] = shift
        return mv @ back @ r @ to
    mid = 0.5 * (lo + hi)
    j1 = about(mid, 0.5 * phase, scale=1.0 + 0.2 * phase)
    j2 = j1 @ about(hi, 0.8 * phase, shift=(0.05 * phase, 0.1 * phase, 0))
    return skins, np.stack([np.eye(4), j1, j2]).astype(np.float32)


@pytest.mark.gpu
def test_skinned_mesh(R, ctx, oracle):
    """shader/skinning.comp + the acceleration-structure update of scene_stage::record_skinning
    (src/scene_stage.cc:1543-1612): skinned vertices bit-equal to the oracle's, then the refitted structure renders what
    an oracle scene built from those vertices renders."""
    from tauray_amd.gltf import load_glb
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), 128, 128)
    ss = R.SceneStage(ctx, scene)
    inst = 4                                                    # the teapot
    sp = scene.spans[inst]
    bind = scene.vertices[sp["vertex_offset"]:sp["vertex_offset"] + sp["vertex_count"]].copy()
    assert np.array_equal(ss.vertices(inst).view(np.uint8), bind.view(np.uint8))
    with pytest.raises(R.TrhipError):
        ss.skin(inst, np.eye(4)[None])                          # no skin set yet
    skins, _ = _bend_rig(bind, 0.0)
    with pytest.raises(R.TrhipError):
        ss.set_skin(inst, skins[:-1])                           # one record per vertex of the mesh
    ss.set_skin(inst, skins)                                    # source = the uploaded vertices

    def feature(fid):
        fs = R.FeatureStage(ctx, ss, fid, _dup((128, 128)))
        buf = ctx.alloc(128 * 128 * 16).zero()
        fs.run(buf)
        return buf.download((128, 128, 4))

    before = feature(3)
    frames = []
    for phase, refit in ((1.0, True), (-0.7, True), (0.4, False)):
        _, joints = _bend_rig(bind, phase)
        info = ss.skin(inst, joints, refit=refit)
        assert info["build_ms"] > 0
        want = oracle.skin_vertices(bind, skins, joints)
        assert np.array_equal(ss.vertices(inst).view(np.uint8), want.view(np.uint8)), f"skinned vertices, phase {phase}"
        osc = oracle.OracleScene(posed_scene(scene, sp, want))
        for fid in (5, 3, 1, 9):      # distance, world position, world normal, instance id
            assert np.array_equal(feature(fid), osc.render_feature(fid, 128, 128)), f"feature {fid}, phase {phase}, refit={refit}"
        img = _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=3)
        _compare(img, osc.render_pt(oracle.options_for_scene(scene, max_bounces=3), 128, 128), f"skinned frame, phase {phase}")
        frames.append(img)
    assert float(np.abs(feature(3) - before)[np.isfinite(before)].max()) > 0.05, "the mesh did not move"
    # the same pose after a rebuild instead of a refit: same frame, bit for bit
    _, joints = _bend_rig(bind, -0.7)
    ss.skin(inst, joints, refit=True)
    a = _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=3)
    assert np.array_equal(a, frames[1])
    ss.skin(inst, joints, refit=False)
    assert np.array_equal(a, _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=3)), "refit vs rebuild"
    # pre-transformed vertices follow the skinned mesh
    osc = oracle.OracleScene(posed_scene(scene, sp, oracle.skin_vertices(bind, skins, joints)))
    got = _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=2, pre_transformed_vertices=1)
    _compare(got, osc.render_pt(oracle.options_for_scene(scene, max_bounces=2, pre_transformed_vertices=1), 128, 128), "skinned + pre-transformed")
    # rendering between trhip_scene_skin and the update is refused
    ss.skin(inst, joints, refit=None)
    with pytest.raises(R.TrhipError):
        _render_hip(R, ctx, ss, scene, (128, 128), max_bounces=1)
    ss._accel_after_change(True)


def posed_scene(scene, sp, vertices):
    import copy
    posed = copy.copy(scene)
    posed.vertices = scene.vertices.copy()
    posed.vertices[sp["vertex_offset"]:sp["vertex_offset"] + sp["vertex_count"]] = vertices
    return posed


@pytest.mark.gpu
def test_view_and_sample_shards(R, ctx, oracle):
    """View and sample sharding: a shard's images are exactly the ones a single device renders for those
    viewports / samples, because cameras and RNG streams are addressed by global viewport and sample index."""
    import copy
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene import generate_camera_grid
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), 96, 64)
    scene = copy.copy(scene)
    scene.cameras = generate_camera_grid(scene.cameras[0], 3, 2, 0.3, 0.3, 5.0)            # 6 views
    ss = R.SceneStage(ctx, scene)
    size, V = (96, 64), 6
    kw = dict(max_bounces=3)
    whole = _render_hip(R, ctx, ss, scene, size, viewports=V, **kw)
    osc = oracle.OracleScene(scene)

    def shard_render(viewports, frames=1, **shard):
        opt = R.options_for_scene(scene, **dict(kw, **shard.pop("opt", {})))
        pt = R.PathTracerStage(ctx, ss, opt, _dup(size))
        pt.set_shard(**shard)
        color = ctx.alloc(viewports * size[0] * size[1] * 16).zero()
        for _ in range(frames):
            pt.run(color, viewports)
        img = color.download((viewports, size[1], size[0], 4))
        pt.close()
        return img

    for world in (2, 4):
        for rank in range(world):
            mine = list(range(rank, V, world))
            got = shard_render(len(mine), viewport_base=rank, viewport_stride=world)
            assert np.array_equal(got, whole[mine]), f"view shard {rank}/{world}"
    osc.set_shard(viewport_base=1, viewport_stride=2)
    _compare(whole[1::2], osc.render_pt(oracle.options_for_scene(scene, **kw), *size, viewports=3), "view shard vs oracle")
    osc.set_shard()
    with pytest.raises(R.TrhipError):
        shard_render(4, viewport_base=1, viewport_stride=2)         # viewport 7 does not exist

    # samples: 8 spp on one device against 4 shards of 2 spp, two frames accumulated (the sample counter advances by the
    # whole job's samples per frame)
    S, world = 8, 4
    one = _render_hip(R, ctx, ss, scene, size, frames=2, samples_per_pixel=S, **kw)
    parts = [shard_render(1, frames=2, sample_base=r, sample_stride=world, opt=dict(samples_per_pixel=S // world)) for r in range(world)]
    mean = np.sum(np.stack(parts).astype(np.float64), axis=0) / world
    assert float(np.abs(mean[..., :3] - one[..., :3]).max()) <= 2e-6 * max(1.0, float(np.abs(one[..., :3]).max())), "sample shards do not add up"
    assert not np.array_equal(parts[0], parts[1])
    # the Sobol samplers address their sequences by the same global sample index
    for sampler in (1, 2, 3):
        one_s = _render_hip(R, ctx, ss, scene, size, samples_per_pixel=4, sampler=sampler, **kw)
        parts_s = [shard_render(1, sample_base=r, sample_stride=2, opt=dict(samples_per_pixel=2, sampler=sampler)) for r in range(2)]
        mean_s = np.sum(np.stack(parts_s).astype(np.float64), axis=0) / 2
        assert float(np.abs(mean_s[..., :3] - one_s[..., :3]).max()) <= 2e-6 * max(1.0, float(np.abs(one_s[..., :3]).max())), f"sampler {sampler}"
        got_v = shard_render(3, viewport_base=1, viewport_stride=2, opt=dict(sampler=sampler))
        assert np.array_equal(got_v, _render_hip(R, ctx, ss, scene, size, viewports=V, sampler=sampler, **kw)[1::2]), f"view shard, sampler {sampler}"
    # one shard against the oracle's restatement of the same shard
    osc.set_shard(sample_base=2, sample_stride=4)
    _compare(shard_render(1, sample_base=2, sample_stride=4, opt=dict(samples_per_pixel=2)),
             osc.render_pt(oracle.options_for_scene(scene, samples_per_pixel=2, **kw), *size), "sample shard vs oracle")
    osc.set_shard()
    with pytest.raises(R.TrhipError):
        shard_render(1, sample_base=4, sample_stride=4)


@pytest.mark.gpu
def test_rt_renderer_view_shard(R, ctx, oracle):
    """RtRenderer(shard="views"): rank r of N renders and tonemaps viewports r, r+N, ... with nothing exchanged."""
    import copy
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene import generate_camera_grid
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), 64, 48)
    scene = copy.copy(scene)
    scene.cameras = generate_camera_grid(scene.cameras[0], 5, 1, 0.3, 0.3, 5.0)
    opt = R.options_for_scene(scene, max_bounces=2)
    single = R.RtRenderer(ctx, scene, opt, (64, 48), viewports=5, use_torch=False)
    single.render()
    want_color, want_display = single.download("color"), single.download("display")
    for world in (2, 3, 7):
        for rank in range(world):
            rr = R.RtRenderer(ctx, scene, opt, (64, 48), rank=rank, world_size=world, viewports=5, use_torch=False, shard="views")
            assert rr.viewports == len(range(rank, 5, world))
            rr.render()
            if rr.viewports:
                assert np.array_equal(rr.download("color"), want_color[rank::world])
                assert np.array_equal(rr.download("display"), want_display[rank::world])
    with pytest.raises(ValueError):
        R.RtRenderer(ctx, scene, R.options_for_scene(scene, samples_per_pixel=3), (64, 48), rank=0, world_size=2, use_torch=False, shard="samples")


@pytest.mark.gpu
def test_frames_in_flight_are_the_same_frames(R, ctx):
    """Frame slots (MAX_FRAMES_IN_FLIGHT, src/context.hh:26): frame i on slot i mod F, started before the previous frames
    have finished, is bit for bit the frame a one-frame-at-a-time renderer produces for index i."""
    from tauray_amd.gltf import load_glb
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), 160, 96)
    opt = R.options_for_scene(scene, max_bounces=3)
    serial = R.RtRenderer(ctx, scene, opt, (160, 96), use_torch=False)
    want = []
    for _ in range(7):
        serial.render()
        want.append((serial.download("color"), serial.download("display")))
    assert not np.array_equal(want[0][0], want[1][0])
    cs = serial.counters()
    serial.close()
    for F in (2, 3):
        rr = R.RtRenderer(ctx, scene, opt, (160, 96), use_torch=False, frames_in_flight=F)
        for i in range(7):          # no host synchronisation between frames
            rr.render()
        rr.sync()
        for back in range(F):       # the last F frames are still in their slots
            i = 6 - back
            slot = rr.slots[i % F]
            assert np.array_equal(slot.color.download((1, 96, 160, 4)), want[i][0]), f"F={F}: frame {i}"
            assert np.array_equal(slot.display.download((1, 96, 160, 4)), want[i][1]), f"F={F}: tonemapped frame {i}"
        c = rr.counters()
        assert c["closest_rays"] == cs["closest_rays"] and c["shadow_rays"] == cs["shadow_rays"] and c["stack_overflows"] == 0
        rr.reset_accumulation(reset_sample_counter=True)
        rr.render()
        assert np.array_equal(rr.download("color"), want[0][0])
        rr.close()
    with pytest.raises(ValueError):
        R.RtRenderer(ctx, scene, opt, (160, 96), use_torch=False, frames_in_flight=2, accumulate=True)
    # rt_renderer<direct_stage>: the same renderer around the other pipeline, one frame at a time and with slots
    dopt = R.options_for_scene(scene, max_bounces=3, samples_per_pixel=2, samples_per_pass=2)
    a = R.RtRenderer(ctx, scene, dopt, (160, 96), use_torch=False, stage_cls=R.DirectStage)
    b = R.RtRenderer(ctx, scene, dopt, (160, 96), use_torch=False, stage_cls=R.DirectStage, frames_in_flight=3)
    for i in range(4):
        a.render(); b.render()
        assert np.array_equal(a.download("display"), b.download("display")), f"direct stage, frame {i}"
    assert not np.array_equal(a.download("color"), want[3][0])
    a.close(); b.close()


@pytest.mark.gpu
def test_skinned_glb_scene(R, ctx, oracle):
    """A glTF file with a skin (tests/golden/skinned.glb): the scene stage poses the mesh with the file's rest pose on upload,
    `pose()` moves the joints; frames equal the oracle's for the same joint transforms."""
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene import trs_matrix
    scene = load_glb(os.path.join(GOLDEN, "skinned.glb"), 128, 128)
    ss = R.SceneStage(ctx, scene)
    sk = scene.skinned[0]
    sp = scene.spans[sk.instance]
    bind = scene.vertices[sp["vertex_offset"]:sp["vertex_offset"] + sp["vertex_count"]]
    assert np.array_equal(ss.vertices(0).view(np.uint8), oracle.skin_vertices(bind, sk.skins, scene.joint_transforms(sk)).view(np.uint8))

    def feature(fid):
        fs = R.FeatureStage(ctx, ss, fid, _dup((128, 128)))
        buf = ctx.alloc(128 * 128 * 16).zero()
        fs.run(buf)
        return buf.download((128, 128, 4))

    def check(osc, what):
        for fid in (5, 3, 1, 9):
            assert np.array_equal(feature(fid), osc.render_feature(fid, 128, 128), equal_nan=True), f"feature {fid}, {what}"   # misses are NaN
        _compare(_render_hip(R, ctx, ss, scene, (128, 128), max_bounces=3),
                 osc.render_pt(oracle.options_for_scene(scene, max_bounces=3), 128, 128), what)

    check(oracle.OracleScene(scene), "rest pose")
    rest_ids = feature(9)[..., 0]
    assert (rest_ids == 0).sum() > 100, "the tube is not in view"
    # an animation step of the caller's: joint 1 swings the other way, joint 2 curls
    g = dict(scene.node_globals)
    g[3] = g[2] @ trs_matrix((0, 1, 0), (0, 0, np.sin(-0.35), np.cos(-0.35)))
    g[4] = g[3] @ trs_matrix((0, 1, 0), (np.sin(0.3), 0, 0, np.cos(0.3)))
    ss.pose(g)
    check(oracle.OracleScene(scene, node_globals=g), "new pose")
    assert (feature(9)[..., 0] != rest_ids).sum() > 50, "the pose did not change the image"
    # an emissive skinned mesh: its triangles are light sources whose records follow the pose (extract_tri_lights.comp runs
    # after skinning.comp in scene_stage::update)
    scene.instances["mat"]["emission_factor"][sk.instance] = (2.0, 1.0, 0.4, 1.0)
    scene.finalize(True)
    ss = R.SceneStage(ctx, scene)
    for pose, refit in ((None, None), (g, True), (g, False)):
        if pose is not None:
            ss.pose(pose, refit=refit)
        osc = oracle.OracleScene(scene, node_globals=pose)
        assert ss.accel["tri_light_count"] == 768
        assert np.array_equal(ss.tri_lights().view(np.uint8), osc.tri_lights().view(np.uint8)), f"tri lights, pose {pose is not None}, refit {refit}"
        _compare(_render_hip(R, ctx, ss, scene, (128, 128), max_bounces=3),
                 osc.render_pt(oracle.options_for_scene(scene, max_bounces=3), 128, 128), "emissive skinned mesh")


@pytest.mark.gpu
def test_tree_optimisation_changes_the_work_not_the_hits(R, ctx, monkeypatch):
    """trhip_scene_set_build_mode: the static build (reinsertion rounds on the binary tree, csrc/bvh_optimize.h) and the fast
    build give the same hits - the same frame bit for bit - and the optimised tree needs fewer node visits for them."""
    from tauray_amd import scenes
    W, H = 640, 360
    scene = scenes.sponza_class(seed=5, target_tris=120000, width=W, height=H)
    opt = R.options_for_scene(scene, max_bounces=3)

    def frame_and_visits(ss):
        pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
        pt.set_profiling(True, False)
        buf = ctx.alloc(W * H * 16).zero()
        pt.run(buf)
        c = pt.counters()
        pt.close()
        assert c["stack_overflows"] == 0
        return buf.download((H, W, 4)), c["node_visits"], c["closest_rays"] + c["shadow_rays"]

    ss = R.SceneStage(ctx, scene)                       # first build of a scene: prefer fast trace
    a, visits_opt, rays = frame_and_visits(ss)
    ss.update_instances(scene.instances)                # a rebuild after a change: prefer fast build
    b, visits_fast, rays_b = frame_and_visits(ss)
    assert np.array_equal(a, b) and rays == rays_b, f"{int((a != b).any(-1).sum())} pixels differ between the optimised and the fast build"
    assert visits_opt < 0.99 * visits_fast, (visits_opt, visits_fast)
    ss2 = R.SceneStage(ctx, scene, fast_trace_rebuilds=True)
    # with TRHIP_DEBUG a static build checks itself after every reinsertion round: every box against its children's, every leaf
    # count, every parent link (k_opt_check); an inconsistent tree fails the build
    monkeypatch.setenv("TRHIP_DEBUG", "1")
    for _ in range(5):
        ss2.update_instances(scene.instances)
    monkeypatch.delenv("TRHIP_DEBUG")
    ss2.update_instances(scene.instances)
    c, visits_again, _ = frame_and_visits(ss2)
    assert np.array_equal(a, c) and abs(visits_again - visits_opt) <= 1e-3 * visits_opt, (visits_again, visits_opt)


@pytest.mark.gpu
def test_sample_lanes_render_the_frame_of_pixel_lanes(R, ctx, oracle):
    """A frame of several one-sample passes keeps whole samples in flight on the stage's lanes (each with path state of its own,
    k_resolve in pass order) instead of slices of one sample: the same bits as four pixel lanes, as one lane, and as the
    oracle's frame within the usual tolerance; also with the demodulated targets, 3 passes on 4 lanes, and accumulated on top of
    an earlier frame."""
    from tauray_amd import scenes
    W, H = 480, 270
    scene = scenes.sponza_class(seed=9, target_tris=50000, width=W, height=H)
    ss = R.SceneStage(ctx, scene)
    for spp, extra in ((8, {}), (3, {}), (5, dict(film=2, sampler=1))):
        opt = R.options_for_scene(scene, max_bounces=3, samples_per_pixel=spp, samples_per_pass=1, **extra)
        frames = {}
        for lanes in (0, 4, 1):
            pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
            pt.set_lanes(lanes)
            color, diffuse, reflection = (ctx.alloc(W * H * 16).zero() for _ in range(3))
            targets = {"color": color, "diffuse": diffuse, "reflection": reflection}
            pt.run_targets(targets)
            pt.run_targets(targets)          # a second frame accumulates on top (samples_accumulated = spp)
            assert pt.counters()["stack_overflows"] == 0
            frames[lanes] = np.concatenate([b.download((H, W, 4)) for b in (color, diffuse, reflection)], axis=-1)
            pt.close()
        assert np.isfinite(frames[0]).all() and frames[0][..., :3].mean() > 1e-3
        for lanes in (4, 1):
            assert np.array_equal(frames[0], frames[lanes]), f"{spp} spp: sample lanes vs {lanes} pixel lane(s): {int((frames[0] != frames[lanes]).any(-1).sum())} pixels differ"
    osc = oracle.OracleScene(scene)
    opt = R.options_for_scene(scene, max_bounces=3, samples_per_pixel=8, samples_per_pass=1)
    _compare(_render_hip(R, ctx, ss, scene, (W, H), max_bounces=3, samples_per_pixel=8, samples_per_pass=1),
             osc.render_pt(oracle.options_for_scene(scene, max_bounces=3, samples_per_pixel=8, samples_per_pass=1), W, H), "8 one-sample passes")


@pytest.mark.gpu
def test_animated_glb_playback(R, ctx, oracle):
    """tests/golden/animated.glb played the way `tauray --animation --framerate 24` plays a file (src/tauray.cc:252-253,
    1052-1092): per frame the animator moves nodes, cameras and joints, SceneStage.animate uploads instance records, joint
    matrices and cameras and updates the acceleration structure (refit, every third frame a rebuild).  Checked frames equal an
    oracle scene built from the same transforms: geometry features bit for bit, motion features and path-traced frames within
    the usual tolerances."""
    from tauray_amd.gltf import load_glb
    from tauray_amd.animation import SceneAnimator
    W = H = 128
    scene = load_glb(os.path.join(GOLDEN, "animated.glb"), W, H)
    ss = R.SceneStage(ctx, scene)
    an = SceneAnimator(scene)
    an.play("")
    dt = round(1000000.0 / 24.0)

    def feature(fid):
        fs = R.FeatureStage(ctx, ss, fid, _dup((W, H)))
        buf = ctx.alloc(W * H * 16).zero()
        fs.run(buf)
        return buf.download((H, W, 4))

    ids = []
    frame = 0
    while True:
        ss.animate(an, 0 if frame == 0 else dt, refit=(frame % 3 != 2))
        if not an.is_playing():
            break
        if frame in (0, 5, 13, 22, 29):
            osc = oracle.OracleScene(scene, node_globals=an.node_globals)
            osc.set_previous_cameras(an.previous_cameras)
            for fid in (5, 3, 1, 9):      # distance, world position, world normal, instance id
                assert np.array_equal(feature(fid), osc.render_feature(fid, W, H), equal_nan=True), f"feature {fid}, frame {frame}"
            if frame > 0:
                hit = np.isfinite(feature(5)[..., 0])
                for fid in (6, 8):        # world motion, screen motion: model_prev and the previous camera are last frame's
                    g, r = feature(fid), osc.render_feature(fid, W, H)
                    assert float(np.abs(g[hit] - r[hit]).max()) <= 1e-5 * max(1.0, float(np.abs(r[hit]).max())), f"motion feature {fid}, frame {frame}"
                assert float(np.abs(feature(8)[hit][..., :2]).max()) > 1e-3, "nothing moved on screen"
            _compare(_render_hip(R, ctx, ss, scene, (W, H), max_bounces=3),
                     osc.render_pt(oracle.options_for_scene(scene, max_bounces=3), W, H), f"animated frame {frame}")
            ids.append(feature(9)[..., 0].copy())
        frame += 1
    assert frame == 30, frame            # 1.25 s at 24 fps: the clip ends when the timer reaches its last key
    assert all((ids[k] != ids[k + 1]).sum() > 50 for k in range(len(ids) - 1)), "the picture did not change between checked frames"


@pytest.mark.gpu
def test_frame_batches_render_the_frames_of_separate_calls(R, ctx):
    """trhip_pt_set_frame_batch / RtRenderer(frames_per_launch=B): B consecutive frames in one launch - frame-major layer groups of
    one image - are the frames separate render calls produce, bit for bit: one viewport and a 3-view camera grid, whole frames and
    a strip shard, and through the renderer with frames in flight (tonemapped layers)."""
    from tauray_amd import scenes
    from tauray_amd import distribution as D
    from tauray_amd.scene import generate_camera_grid
    W, H = 320, 180
    scene = scenes.sponza_class(seed=3, target_tris=30000, width=W, height=H)
    scene.cameras = generate_camera_grid(scene.cameras[0], 3, 1, 0.2, 0.2, 5.0)
    ss = R.SceneStage(ctx, scene)
    opt = R.options_for_scene(scene, max_bounces=3)
    for views, dist in ((1, _dup((W, H))), (3, _dup((W, H))), (1, D.get_device_distribution_params((W, H), D.DISTRIBUTION_SHUFFLED_STRIPS, 0.3, 0.45, 1, 3, False))):
        tw, th = D.get_distribution_target_size(dist)
        single = []
        pt = R.PathTracerStage(ctx, ss, opt, dist)
        for f in range(6):
            buf = ctx.alloc(views * tw * th * 16).zero()
            pt.reset_accumulated_samples()
            pt.run(buf, views)
            single.append(buf.download((views, th, tw, 4)))
        pt.close()
        assert not np.array_equal(single[0], single[1])
        for B in (2, 3):
            pt = R.PathTracerStage(ctx, ss, opt, dist)
            pt.set_frame_batch(B)
            for call in range(6 // B):
                buf = ctx.alloc(B * views * tw * th * 16).zero()
                pt.reset_accumulated_samples()
                pt.run(buf, B * views)
                got = buf.download((B, views, th, tw, 4))
                for k in range(B):
                    assert np.array_equal(got[k], single[call * B + k]), f"{views} view(s), batch of {B}: frame {call * B + k} differs in {int((got[k] != single[call * B + k]).any(-1).sum())} pixels"
            assert pt.counters()["stack_overflows"] == 0
            with pytest.raises(R.TrhipError):
                pt.run(buf, B * views + 1)           # not a whole number of frames
            pt.close()
    # the renderer: six frames as two launches of three, two launches in flight
    one = R.RtRenderer(ctx, scene, opt, (W, H), viewports=3)
    ref = []
    for f in range(6):
        one.reset_accumulation()
        one.render()
        ref.append(one.download("display"))
    one.close()
    rr = R.RtRenderer(ctx, scene, opt, (W, H), viewports=3, frames_in_flight=2, frames_per_launch=3)
    for call in range(2):
        rr.render()
        got = rr.download("display").reshape(3, 3, H, W, 4)
        for k in range(3):
            assert np.array_equal(got[k], ref[call * 3 + k]), f"renderer, frame {call * 3 + k}"
    rr.close()


@pytest.mark.gpu
def test_full_size_properties_one_million_triangles(R, ctx, monkeypatch):
    """BASELINE config 4 at its full size (sponza_teapots, 1920x1080, 4 bounces), through properties that need no oracle: the
    frame does not depend on the tree (PLOC vs LBVH build, refit vs rebuild), on how it is sharded (8 shuffled-strip shards,
    stitched in one launch) or on how many frames are in flight."""
    from tauray_amd import scenes
    from tauray_amd import distribution as D
    W, H = 1920, 1080
    scene = scenes.sponza_teapots(W, H)
    ss = R.SceneStage(ctx, scene)
    assert ss.accel["triangle_count"] > 900_000
    a = _render_hip(R, ctx, ss, scene, (W, H), max_bounces=4)
    assert np.isfinite(a).all() and (a[..., 3] == 1).all() and a[..., :3].min() >= 0 and a[..., :3].mean() > 1e-3
    assert ss.accel["leaf_count"] == ss.accel["triangle_count"] and ss.accel["node_count"] == ss.accel["leaf_count"] - 1
    # another tree over the same triangles (a rebuild is a fast build: no references, no optimisation rounds): every hit is the
    # same hit, so the frame is bit-identical
    monkeypatch.setenv("TRHIP_BUILDER", "lbvh")
    ss.update_instances(scene.instances)
    assert ss.accel["leaf_count"] == ss.accel["triangle_count"]
    b = _render_hip(R, ctx, ss, scene, (W, H), max_bounces=4)
    assert np.array_equal(b, a), f"LBVH tree vs PLOC tree with split triangles: {int((b != a).any(-1).sum())} pixels differ"
    monkeypatch.setenv("TRHIP_BUILDER", "ploc")
    ss.update_instances(scene.instances)
    assert ss.accel["node_count"] == ss.accel["triangle_count"] - 1
    ss.update_instances(scene.instances, refit=True)
    assert np.array_equal(_render_hip(R, ctx, ss, scene, (W, H), max_bounces=4), a), "refitted tree"
    # 8 shuffled-strip shards, all partials stitched by one launch
    opt = R.options_for_scene(scene, max_bounces=4)
    primary = ctx.alloc(W * H * 16).zero()
    dists = [D.get_device_distribution_params((W, H), D.DISTRIBUTION_SHUFFLED_STRIPS, i / 8, 1 / 8, i, 8, i == 0) for i in range(8)]
    parts = []
    for i, d in enumerate(dists):
        pt = R.PathTracerStage(ctx, ss, opt, d)
        if i == 0:
            pt.run(primary)
        else:
            tw, th = D.get_distribution_target_size(d)
            parts.append(ctx.alloc(tw * th * 16).zero())
            pt.run(parts[-1])
        pt.close()
    R.StitchStage(ctx, (W, H)).run_all(dists[1:], parts, primary)
    assert np.array_equal(primary.download((1, H, W, 4)), a), "8 strip shards"
    del parts
    # one lane or two instead of the automatic four concurrent slices of the frame
    for lanes in (1, 2):
        pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
        pt.set_lanes(lanes)
        buf = ctx.alloc(W * H * 16).zero()
        pt.run(buf)
        assert np.array_equal(buf.download((1, H, W, 4)), a), f"{lanes} lane(s)"
        pt.close()
    # three frames in flight: frame 0 of the slots is the frame above, frame 2 equals a serial renderer's frame 2
    pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
    serial = ctx.alloc(W * H * 16).zero()
    for _ in range(3):
        pt.reset_accumulated_samples()
        pt.run(serial)
    frame2 = serial.download((1, H, W, 4))
    pt.close()
    rr = R.RtRenderer(ctx, scene, opt, (W, H), use_torch=False, frames_in_flight=3)     # uploads the same scene again
    for _ in range(3):
        rr.render()
    rr.sync()
    assert np.array_equal(rr.slots[0].color.download((1, H, W, 4)), a) and np.array_equal(rr.slots[2].color.download((1, H, W, 4)), frame2)
    rr.close()


@pytest.mark.gpu
def test_maximum_size_frame(R, ctx):
    """An 8K frame (7680x4320: 33 M paths, ~8 GB of path state on a 288 GB device): deterministic, and two scanline shards
    stitched together give the same bits; 32-bit path ids and queue counters have room for 128 such frames."""
    from tauray_amd import scenes
    from tauray_amd import distribution as D
    W, H = 7680, 4320
    scene = scenes.test_glb(W, H)
    ss = R.SceneStage(ctx, scene)
    kw = dict(max_bounces=2)
    a = _render_hip(R, ctx, ss, scene, (W, H), **kw)
    assert a.shape == (1, H, W, 4) and np.isfinite(a).all() and (a[..., 3] == 1).all() and a[..., :3].mean() > 1e-3
    opt = R.options_for_scene(scene, **kw)
    dists = [D.get_device_distribution_params((W, H), D.DISTRIBUTION_SCANLINE, i / 2, 1 / 2, i, 2, i == 0) for i in range(2)]
    primary = ctx.alloc(W * H * 16).zero()
    pt = R.PathTracerStage(ctx, ss, opt, dists[0]); pt.run(primary); pt.close()
    tw, th = D.get_distribution_target_size(dists[1])
    part = ctx.alloc(tw * th * 16).zero()
    pt = R.PathTracerStage(ctx, ss, opt, dists[1]); pt.run(part); pt.close()
    R.StitchStage(ctx, (W, H)).run_all(dists[1:], [part], primary)
    assert np.array_equal(primary.download((1, H, W, 4)), a)


@pytest.mark.gpu
def test_long_accumulation_is_the_mean_of_its_frames(R, ctx):
    """BASELINE config 3 accumulates 4096 samples per pixel: the running mean of gbuffer.glsl:18-28 over many frames must stay
    the mean of the frames.  512 accumulated 1-spp frames against the float64 mean of the same 512 frames rendered one by one."""
    from tauray_amd.gltf import load_glb
    W, H, N = 48, 32, 512
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    ss = R.SceneStage(ctx, scene)
    opt = R.options_for_scene(scene, max_bounces=3)
    pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
    acc = ctx.alloc(W * H * 16).zero()
    for _ in range(N):
        pt.run(acc)
    running = acc.download((1, H, W, 4)).astype(np.float64)
    pt.reset_accumulated_samples(); pt.reset_sample_counter()
    total = np.zeros((1, H, W, 4))
    for _ in range(N):
        pt.reset_accumulated_samples()
        pt.run(acc)
        total += acc.download((1, H, W, 4))
    pt.close()
    mean = total / N
    assert np.abs(running[..., :3] - mean[..., :3]).max() <= 2e-5 * max(1.0, mean[..., :3].max())
    assert (running[..., 3] == 1).all()


@pytest.mark.gpu
def test_sharded_gbuffer_direct_and_feature_stages(R, ctx):
    """Pixel sharding of everything that renders: a scanline / strip shard of the gbuffer targets, of direct_stage and of
    feature_stage holds exactly the pixels the unsharded image has at those positions (every stage keys its random streams
    and its write position by the absolute pixel, shader/rt.glsl:170-231)."""
    from tauray_amd import distribution as D
    from tauray_amd.gltf import load_glb
    W, H = 256, 144
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    ss = R.SceneStage(ctx, scene)
    names = ["color", "diffuse", "reflection", "albedo", "normal", "pos", "instance_id"]
    opt = R.options_for_scene(scene, max_bounces=3)

    def targets(stage_cls, d, options):
        pt = stage_cls(ctx, ss, options, d)
        tw, th = D.get_distribution_target_size(d)
        bufs = {n: ctx.alloc(tw * th * R.PathTracerStage.TARGETS[n][0] * 4).zero() for n in names}
        pt.run_targets(bufs)
        out = {n: np.frombuffer(bufs[n].download((1, th, tw, R.PathTracerStage.TARGETS[n][0])).tobytes(),
                                dtype=R.PathTracerStage.TARGETS[n][1]).reshape(th, tw, -1) for n in names}
        pt.close()
        return out

    def positions(d):
        """absolute pixel of every element of a non-primary shard's target, row-major"""
        tw, th = D.get_distribution_target_size(d)
        if d.strategy == D.DISTRIBUTION_SCANLINE:
            ys = np.arange(th) * d.count + d.index
            return np.repeat(ys, tw), np.tile(np.arange(tw), th), np.repeat(ys < H, tw)
        b = D.calculate_shuffled_strips_b((W, H))
        p = np.arange(tw * th)
        j = np.array([D.permute_region_id(d.index + int(q), d.size, b) if q < d.count else W * H for q in p])
        ok = j < W * H
        j = np.where(ok, j, 0)
        return j // W, j % W, ok

    dopt = R.options_for_scene(scene, max_bounces=3, samples_per_pixel=2, samples_per_pass=2)
    for stage_cls, options, what in ((R.PathTracerStage, opt, "path tracer"), (R.DirectStage, dopt, "direct stage")):
        full = targets(stage_cls, _dup((W, H)), options)
        for strategy, world in ((D.DISTRIBUTION_SCANLINE, 3), (D.DISTRIBUTION_SHUFFLED_STRIPS, 4)):
            for rank in range(1, world):
                d = D.get_device_distribution_params((W, H), strategy, rank / world, 1 / world, rank, world, False)
                part = targets(stage_cls, d, options)
                ys, xs, ok = positions(d)
                for n in names:
                    got = part[n].reshape(-1, part[n].shape[-1])[ok]
                    want = full[n][ys[ok], xs[ok]]
                    assert np.array_equal(got, want, equal_nan=True), f"{what}, strategy {strategy}, rank {rank}/{world}, target {n}"
    # feature_stage the same way
    for fid in (3, 9):
        fs = R.FeatureStage(ctx, ss, fid, _dup((W, H)))
        buf = ctx.alloc(W * H * 16).zero()
        fs.run(buf)
        full = buf.download((H, W, 4))
        d = D.get_device_distribution_params((W, H), D.DISTRIBUTION_SCANLINE, 2 / 3, 1 / 3, 2, 3, False)
        tw, th = D.get_distribution_target_size(d)
        fs = R.FeatureStage(ctx, ss, fid, d)
        buf = ctx.alloc(tw * th * 16).zero()
        fs.run(buf)
        part = buf.download((th, tw, 4))
        ys, xs, ok = positions(d)
        assert np.array_equal(part.reshape(-1, 4)[ok], full[ys[ok], xs[ok]], equal_nan=True), f"feature {fid} shard"


@pytest.mark.gpu
def test_ragged_shards(R, ctx):
    """More devices than scanlines, strips that do not divide the image, one-pixel frames: shards may be empty or ragged
    and the stitched frame still equals the unsharded one."""
    from tauray_amd import distribution as D
    from tauray_amd.gltf import load_glb
    for (W, H), world, strategy in (((16, 5), 8, D.DISTRIBUTION_SCANLINE), ((37, 11), 7, D.DISTRIBUTION_SHUFFLED_STRIPS), ((1, 1), 3, D.DISTRIBUTION_SCANLINE),
                                    ((130, 3), 5, D.DISTRIBUTION_SHUFFLED_STRIPS)):
        scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
        ss = R.SceneStage(ctx, scene)
        full = _render_hip(R, ctx, ss, scene, (W, H), max_bounces=2)
        opt = R.options_for_scene(scene, max_bounces=2)
        dists, cum = [], 0.0
        for i in range(world):
            dists.append(D.get_device_distribution_params((W, H), strategy, cum, 1.0 / world, i, world, i == 0))
            cum += 1.0 / world
        primary = ctx.alloc(W * H * 16).zero()
        parts, pd = [], []
        for i, d in enumerate(dists):
            tw, th = D.get_distribution_target_size(d)
            pt = R.PathTracerStage(ctx, ss, opt, d)
            if i == 0:
                pt.run(primary)
            else:
                buf = ctx.alloc(max(tw * th, 1) * 16).zero()
                if tw * th > 0:
                    pt.run(buf)
                parts.append(buf); pd.append(d)
            pt.close()
        R.StitchStage(ctx, (W, H)).run_all(pd, parts, primary)
        assert np.array_equal(primary.download((1, H, W, 4)), full), f"{W}x{H} over {world} devices, strategy {strategy}"
        # the renderer object of the last rank (possibly an empty shard) renders without the exchange
        rr = R.RtRenderer(ctx, scene, opt, (W, H), strategy=strategy, rank=world - 1, world_size=world, use_torch=False)
        rr.render_partial()
        rr.sync()
        rr.close()


@pytest.mark.gpu
def test_random_shard_geometries(R, ctx):
    """Seeded draws of what a multi-GPU job can look like: frame sizes that are not multiples of anything, 1-12 devices, scanlines or
    shuffled strips, equal shares or the uneven ones a load balancer hands out (some of them empty), one or several viewports:
    every device's partial frame rendered on this GPU and stitched (trhip_stitch_batch) must equal the unsharded frame bit for
    bit.  TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS_SMALL run longer campaigns (tools/fuzz_campaign.sh)."""
    import copy
    from tauray_amd import distribution as D
    from tauray_amd import scene as S
    from tauray_amd.gltf import load_glb
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "9")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "16"))):
        W, H = int(rng.integers(1, 200)), int(rng.integers(1, 130))
        if k % 4 == 0:
            W = int(rng.choice([8, 64, 128, 136]))            # whole 8x8 tiles in x: the tiled launch order (csrc/shading.h launch_coord)
        world = int(rng.integers(1, 13))
        strategy = int(rng.choice([D.DISTRIBUTION_SCANLINE, D.DISTRIBUTION_SHUFFLED_STRIPS]))
        views = int(rng.choice([1, 1, 1, 3]))
        shares = np.full(world, 1.0 / world)
        if rng.uniform() < 0.5:                               # a load balancer's shares, zeros included
            shares = rng.uniform(0, 1, world) * (rng.uniform(0, 1, world) > 0.2)
            shares = shares / shares.sum() if shares.sum() > 0 else np.full(world, 1.0 / world)
        scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
        if views > 1:
            scene = copy.copy(scene)
            scene.cameras = S.generate_camera_grid(scene.cameras[0], views, 1, 0.3, 0.3, 5.0)
        ss = R.SceneStage(ctx, scene)
        kw = dict(max_bounces=int(rng.integers(1, 4)), sampler=int(rng.integers(0, 4)))
        what = f"draw {k}: {W}x{H}, {views} view(s), {world} devices, strategy {strategy}, shares {np.round(shares, 3).tolist()}, {kw}"
        full = _render_hip(R, ctx, ss, scene, (W, H), viewports=views, **kw)
        opt = R.options_for_scene(scene, **kw)
        dists, cum = [], 0.0
        for i in range(world):
            dists.append(D.get_device_distribution_params((W, H), strategy, cum, float(shares[i]), i, world, i == 0))
            cum += float(shares[i])
        primary = ctx.alloc(W * H * 16 * views).zero()
        parts, pd = [], []
        for i, d in enumerate(dists):
            tw, th = D.get_distribution_target_size(d)
            pt = R.PathTracerStage(ctx, ss, opt, d)
            if i == 0:
                pt.run(primary, views)
            else:
                buf = ctx.alloc(max(tw * th * views, 1) * 16).zero()
                if tw * th > 0:
                    pt.run(buf, views)
                parts.append(buf); pd.append(d)
            pt.close()
        R.StitchStage(ctx, (W, H)).run_all(pd, parts, primary, viewports=views)
        got = primary.download((views, H, W, 4))
        assert np.array_equal(got, full), f"{what}: {(got != full).any(-1).sum()} pixels differ"


@pytest.mark.gpu
def test_texture_edge_cases(R, ctx, oracle):
    """Textures that are not powers of two (3x5, 1x1, 7x2), texture coordinates far outside [0, 1] (repeat wrapping,
    negative values), all four material textures at once (albedo with alpha, metallic-roughness, normal map, emission)."""
    from tauray_amd import scene as S
    rng = np.random.default_rng(5)
    texs = [rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) for (w, h) in ((3, 5), (1, 1), (7, 2), (4, 4))]
    texs[0][..., 3] = rng.integers(0, 2, size=(5, 3)) * 255        # alpha-tested albedo
    texs[2][..., 2] = 255                                          # a normal map pointing mostly outwards
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
    sc = S.SceneDesc(instances=np.concatenate([S.make_instance(np.eye(4), mat), S.make_instance(np.eye(4), plain)]),
                     spans=np.array([(0, 4, 0, 2), (4, 4, 6, 2)], dtype=S.MESH_SPAN), vertices=np.concatenate([quad, back]),
                     indices=np.array([0, 1, 2, 0, 2, 3] * 2, dtype=np.uint32), point_lights=S.make_point_light((20, 20, 20), (0.5, 0.8, 2.5), 0.1),
                     textures=texs, cameras=[cam]).finalize(True)
    ss = R.SceneStage(ctx, sc)
    osc = oracle.OracleScene(sc)
    for fid, tol in ((9, 0.0), (5, 0.0), (1, 0.0), (0, 1e-6)):   # instance id (alpha test picks front or back), distance, normal, albedo
        fs = R.FeatureStage(ctx, ss, fid, _dup((96, 96)))
        buf = ctx.alloc(96 * 96 * 16).zero()
        fs.run(buf)
        g, r = buf.download((96, 96, 4)), osc.render_feature(fid, 96, 96)
        if tol == 0.0:
            assert np.array_equal(g, r, equal_nan=True), f"feature {fid}"
        else:
            assert np.array_equal(np.isnan(g), np.isnan(r)) and np.nanmax(np.abs(g - r)) <= tol, f"feature {fid}"
    ids = osc.render_feature(9, 96, 96)[..., 0]
    # the feature renderer's any-hit uses a fixed cutoff of 1e-4 (shader/rt_feature.rahit:17): only where the filtered alpha is
    # exactly zero does the back quad show
    assert (ids == 0).sum() > 500 and (ids == 1).sum() > 100, "the alpha-tested texture should show both quads"
    # A random normal map puts the view direction within 1e-5 of the mapped normal's horizon for a percent of the paths
    # (view_to_tangent_space clamps z at 1e-5, shader/math.glsl:472-478): there dot(view, normal) is all cancellation, one ulp of its
    # inputs moves the BSDF by a percent, and the two arithmetic modes of k_shade (IEEE fp32 like the oracle / what Vulkan asks of
    # the reference's GLSL, csrc/shade_fast.hip) part on those pixels - as two Vulkan drivers would.  IEEE: the usual bound;
    # default arithmetic: the same image up to 3 % of such pixels, mean radiance equal.
    ref = osc.render_pt(oracle.options_for_scene(sc, max_bounces=3), 192, 192)
    img = _render_hip(R, ctx, ss, sc, (192, 192), max_bounces=3, ieee=True)
    rel = np.abs(img[..., :3] - ref[..., :3]) / (np.abs(ref[..., :3]) + 1e-2)
    print("textured quad: pixels outside tolerance", float((rel.max(-1) > REL_TOL).mean()), "bit-equal", float((img == ref).all(-1).mean()))
    _compare(img, ref, "textured quad, IEEE shading")
    _compare(_render_hip(R, ctx, ss, sc, (192, 192), max_bounces=3), ref, "textured quad, default shading arithmetic", max_bad=0.03)
    got = _render_targets_hip(R, ctx, ss, sc, (96, 96), ["albedo", "material", "normal"], max_bounces=2)
    ref = osc.render_pt_targets(oracle.options_for_scene(sc, max_bounces=2), 96, 96, ["albedo", "material", "normal"])
    for n in ("albedo", "material", "normal"):
        assert np.allclose(got[n], ref[n], atol=2e-6, equal_nan=True), n


def _zoo_scene(materials=None, point=(0, 1, 2), directional=(0, 1), env=True):
    """Corner values of the material model and every light class at once (see test_material_and_light_zoo)."""
    from tauray_amd import scene as S
    quad = np.zeros(4, dtype=S.VERTEX)
    quad["pos"] = [(-0.45, -0.45, 0), (0.45, -0.45, 0), (0.45, 0.45, 0), (-0.45, 0.45, 0)]
    quad["normal"] = (0, 0, 1)
    quad["tangent"] = (1, 0, 0, 1)
    quad["uv"] = [(0, 0), (1, 0), (1, 1), (0, 1)]
    mats = [
        S.make_material(albedo=(0.9, 0.9, 0.9, 1), metallic=1.0, roughness=0.0),                       # mirror
        S.make_material(albedo=(0.9, 0.6, 0.2, 1), metallic=1.0, roughness=0.35),
        # (ior exactly 1.0 is left out: transmission then divides by (eta * cos_d + cos_o)^2 = 0 and the reference itself yields
        # NaN, shader/ggx.glsl:350-372)
        S.make_material(albedo=(1, 1, 1, 1), metallic=0.0, roughness=0.0, transmittance=1.0, ior=1.05, double_sided=True),
        S.make_material(albedo=(0.9, 1, 0.9, 1), metallic=0.0, roughness=0.0, transmittance=1.0, ior=0.7, double_sided=True),
        S.make_material(albedo=(1, 0.9, 0.9, 1), metallic=0.0, roughness=0.2, transmittance=0.8, ior=2.4, double_sided=True),
        S.make_material(albedo=(0, 0, 0, 1), metallic=0.0, roughness=1.0),
        S.make_material(albedo=(1, 1, 1, 1), metallic=0.0, roughness=1.0),
        S.make_material(albedo=(0.2, 0.4, 0.9, 1), metallic=0.5, roughness=0.001),                     # just above the delta threshold
        S.make_material(albedo=(0.7, 0.7, 0.2, 0.5), metallic=0.0, roughness=0.5, double_sided=True),   # half-transparent by albedo alpha
    ]
    if materials is not None:
        mats = [mats[k] for k in materials]
    insts, verts, spans, idx = [], [], [], []
    for i, m in enumerate(mats):
        x, y = (i % 3 - 1) * 1.0, (i // 3 - 1) * 1.0
        flip = np.diag([1.0, 1.0, -1.0, 1.0]) if i in (5, 6) else np.eye(4)     # two single-sided panels face away from the camera
        t = S.trs_matrix((x, y, 0.1 * (i % 2))) @ flip
        insts.append(S.make_instance(t, m))
        spans.append((4 * i, 4, 6 * i, 2))
        verts.append(quad)
        idx += [0, 1, 2, 0, 2, 3]
    back = quad.copy(); back["pos"] *= 8
    insts.append(S.make_instance(S.trs_matrix((0, 0, -1.5)), S.make_material(albedo=(0.6, 0.6, 0.6, 1), metallic=0.0, roughness=0.8)))
    spans.append((4 * len(mats), 4, 6 * len(mats), 2)); verts.append(back); idx += [0, 1, 2, 0, 2, 3]
    lights = np.concatenate([
        S.make_point_light((30, 30, 30), (0.3, 0.4, 2.0), 0.0),                                   # radius 0: a delta light
        S.make_point_light((10, 6, 3), (-1.2, 1.0, 1.2), 0.25),                                   # a sphere the camera can see
        S.make_spotlight((40, 40, 60), (1.5, -1.0, 2.5), (-0.5, 0.3, -1.0), 0.05, 25.0, 3.0),
    ])
    dls = np.concatenate([S.make_directional_light((2.0, 1.9, 1.7), (-0.3, -0.5, -1.0), 0.0),     # angle 0: never hit directly
                          S.make_directional_light((0.5, 0.6, 0.9), (0.4, -0.2, -1.0), 5.0)])
    cam = S.Camera(fov=55, aspect=1.0)
    cam.transform = S.trs_matrix((0.15, 0.1, 4.2))
    lights = lights[list(point)] if len(point) else np.zeros(0, dtype=S.POINT_LIGHT)
    dls = dls[list(directional)] if len(directional) else np.zeros(0, dtype=S.DIRECTIONAL_LIGHT)
    return S.SceneDesc(instances=np.concatenate(insts), spans=np.array(spans, dtype=S.MESH_SPAN), vertices=np.concatenate(verts),
                       indices=np.array(idx, dtype=np.uint32), point_lights=lights, directional_lights=dls,
                       envmap=np.ones((2, 4, 4), dtype=np.float32) if env else None,      # a uniform environment: a constant map times a factor
                       environment_factor=(0.3, 0.35, 0.4, 1.0) if env else (0, 0, 0, 0), cameras=[cam]).finalize(True)


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", [0, 1])
def test_material_and_light_zoo(R, ctx, oracle, sampler):
    """Corner values of the material model and every light class at once: mirror (roughness 0), rough metal, glass with
    ior 1.05 / 0.7 / 2.4 and roughness 0, black and white diffuse, single-sided panels seen from behind, a radius-0 point
    light, a spotlight, directional lights of angle 0 and 5 degrees, a uniform environment, a lit sphere light in view."""
    sc = _zoo_scene()
    ss = R.SceneStage(ctx, sc)
    osc = oracle.OracleScene(sc)
    for kw in (dict(max_bounces=6, sampler=sampler), dict(max_bounces=4, sampler=sampler, samples_per_pixel=2, regularization_gamma=0.5)):
        img = _render_hip(R, ctx, ss, sc, (160, 160), **kw)
        ref = osc.render_pt(oracle.options_for_scene(sc, **kw), 160, 160)
        assert np.isfinite(ref).all(), "the oracle produced a non-finite pixel"
        _compare(img, ref, f"zoo {kw}")
    got = _render_targets_hip(R, ctx, ss, sc, (160, 160), ["material", "albedo", "instance_id"], max_bounces=2)
    want = osc.render_pt_targets(oracle.options_for_scene(sc, max_bounces=2), 160, 160, ["material", "albedo", "instance_id"])
    assert np.array_equal(got["instance_id"], want["instance_id"])
    assert np.allclose(got["material"], want["material"], atol=1e-6) and np.allclose(got["albedo"], want["albedo"], atol=1e-6)


@pytest.mark.gpu
def test_odd_frame_sizes_and_scales(R, ctx, oracle):
    """Frame sizes that are not whole 8x8 tiles, with and without concurrent lanes; extreme aspect ratios; the same scene a
    thousand times larger and a thousand times smaller (hit distances scale, the 1e-4 ray offset does not)."""
    import copy
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene import from_glm, to_glm
    # 2.08 M paths: four lanes by default, but 1923 x 1081 is not a whole number of tiles
    W, H = 1923, 1081
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    ss = R.SceneStage(ctx, scene)
    opt = R.options_for_scene(scene, max_bounces=3)
    frames = []
    for lanes in (0, 1, 3):
        pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
        pt.set_lanes(lanes)
        buf = ctx.alloc(W * H * 16).zero()
        pt.run(buf)
        frames.append(buf.download((1, H, W, 4)))
        pt.close()
    assert np.isfinite(frames[0]).all() and np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])
    for (w, h) in ((1920, 1), (1, 1080), (7, 1300), (65, 63)):
        sc = load_glb(os.path.join(GOLDEN, "test.glb"), w, h)
        ss = R.SceneStage(ctx, sc)
        img = _render_hip(R, ctx, ss, sc, (w, h), max_bounces=2)
        if w * h <= 65 * 63:
            _compare(img, oracle.OracleScene(sc).render_pt(oracle.options_for_scene(sc, max_bounces=2), w, h), f"{w}x{h}")
        else:
            assert img.shape == (1, h, w, 4) and np.isfinite(img).all()
    # scale the world (instances, lights, camera) by k: distances and positions scale, everything else stays
    base = load_glb(os.path.join(GOLDEN, "test.glb"), 96, 96)
    for k in (1e3, 1e-3):
        sc = copy.copy(base)
        sc.instances = base.instances.copy()
        S = np.diag([k, k, k, 1.0])
        for i in range(len(sc.instances)):
            m = S @ from_glm(base.instances["model"][i])
            sc.instances["model"][i] = to_glm(m)
            sc.instances["model_prev"][i] = to_glm(m)
            sc.instances["model_normal"][i] = to_glm(np.linalg.inv(m).T)
        sc.point_lights = base.point_lights.copy()
        sc.point_lights["pos"] *= k
        sc.point_lights["radius"] *= k
        sc.point_lights["color"] *= k * k
        sc.cameras = copy.deepcopy(base.cameras)
        for c in sc.cameras:
            t = np.asarray(c.transform, dtype=np.float64).copy()
            t[:3, 3] *= k
            c.transform = t
            c.near *= k; c.far *= k
        ss = R.SceneStage(ctx, sc)
        osc = oracle.OracleScene(sc)
        for fid in (5, 9, 3):
            fs = R.FeatureStage(ctx, ss, fid, _dup((96, 96)), min_ray_dist=1e-4 * min(k, 1.0))
            buf = ctx.alloc(96 * 96 * 16).zero()
            fs.run(buf)
            assert np.array_equal(buf.download((96, 96, 4)), osc.render_feature(fid, 96, 96, min_ray_dist=1e-4 * min(k, 1.0)), equal_nan=True), f"scale {k}, feature {fid}"
        kw = dict(max_bounces=3, min_ray_dist=1e-4 * k)
        _compare(_render_hip(R, ctx, ss, sc, (96, 96), **kw), osc.render_pt(oracle.options_for_scene(sc, **kw), 96, 96), f"scale {k}")


@pytest.mark.gpu
def test_many_instances_of_one_mesh(R, ctx, oracle):
    """4 096 instances that share one vertex / index span (instancing), every one with its own transform and material, a few
    of them emissive: instance lookup, per-instance tri-light ranges and the per-instance pre-transformed copy."""
    from tauray_amd import scene as S
    rng = np.random.default_rng(9)
    n = 4096
    tet = np.zeros(4, dtype=S.VERTEX)
    tet["pos"] = [(0, 0, 0.15), (0.14, 0, -0.07), (-0.07, 0.12, -0.07), (-0.07, -0.12, -0.07)]
    nn = tet["pos"] / np.linalg.norm(tet["pos"], axis=1, keepdims=True)
    tet["normal"] = nn
    tet["tangent"] = (1, 0, 0, 1)
    idx = np.array([0, 1, 2, 0, 2, 3, 0, 3, 1, 1, 3, 2], dtype=np.uint32)
    insts = []
    for i in range(n):
        g = np.array([i % 16, (i // 16) % 16, i // 256], dtype=np.float64)
        t = S.trs_matrix((g - (7.5, 7.5, 7.5)) * 0.45 + rng.uniform(-0.05, 0.05, 3), rng.normal(size=4), rng.uniform(0.6, 1.4, 3))
        emis = (3.0, 2.0, 1.0) if i % 97 == 0 else (0, 0, 0)
        insts.append(S.make_instance(t, S.make_material(albedo=tuple(rng.uniform(0.2, 0.9, 3)) + (1.0,), metallic=float(i % 3 == 0), roughness=float(rng.uniform(0.1, 1.0)),
                                                         emission=emis, double_sided=True)))
    cam = S.Camera(fov=50, aspect=1.0)
    cam.transform = S.trs_matrix((0.3, 0.2, 9.0))
    sc = S.SceneDesc(instances=np.concatenate(insts), spans=np.array([(0, 4, 0, 4)] * n, dtype=S.MESH_SPAN), vertices=tet, indices=idx,
                     point_lights=S.make_point_light((300, 300, 300), (0, 6, 8), 0.3), cameras=[cam]).finalize(True)
    assert sc.triangle_count == 4 * n
    ss = R.SceneStage(ctx, sc)
    osc = oracle.OracleScene(sc)
    assert ss.accel["tri_light_count"] == 4 * len(range(0, n, 97))
    assert np.array_equal(ss.tri_lights().view(np.uint8), osc.tri_lights().view(np.uint8))
    for fid in (9, 5, 1):
        fs = R.FeatureStage(ctx, ss, fid, _dup((128, 128)))
        buf = ctx.alloc(128 * 128 * 16).zero()
        fs.run(buf)
        assert np.array_equal(buf.download((128, 128, 4)), osc.render_feature(fid, 128, 128), equal_nan=True), f"feature {fid}"
    for kw in (dict(max_bounces=3), dict(max_bounces=3, pre_transformed_vertices=1)):
        _compare(_render_hip(R, ctx, ss, sc, (128, 128), **kw), osc.render_pt(oracle.options_for_scene(sc, **kw), 128, 128), f"instanced {kw}")


@pytest.mark.gpu
def test_all_features_on_the_zoo_scene(R, ctx, oracle):
    """feature_stage's ten features on a scene with misses, mirrored instances, single-sided back faces, alpha and a previous
    camera: equal to the oracle's (albedo within the powf difference of the two maths libraries)."""
    import copy
    sc = _zoo_scene()
    ss = R.SceneStage(ctx, sc)
    osc = oracle.OracleScene(sc)
    prev = copy.deepcopy(sc.cameras)
    prev[0].transform = np.asarray(prev[0].transform) @ np.array([[1, 0, 0, 0.1], [0, 1, 0, 0.05], [0, 0, 1, -0.2], [0, 0, 0, 1.0]])
    ss.set_previous_cameras(prev)
    osc.set_previous_cameras(prev)
    for fid in range(10):
        for default in ((np.nan,) * 4, (0.0, -1.0, 2.0, 7.0)):
            fs = R.FeatureStage(ctx, ss, fid, _dup((144, 112)), default_value=default)
            buf = ctx.alloc(144 * 112 * 16).zero()
            fs.run(buf)
            g, r = buf.download((112, 144, 4)), osc.render_feature(fid, 144, 112, default_value=default)
            assert np.array_equal(np.isnan(g), np.isnan(r)), f"feature {fid}"
            tol = 1e-6 if fid in (0, 6, 7, 8) else 0.0
            assert np.nanmax(np.abs(g - r)) <= tol * max(1.0, float(np.nanmax(np.abs(r)))), f"feature {fid}: {np.nanmax(np.abs(g - r))}"


@pytest.mark.gpu
def test_random_option_combinations(R, ctx, oracle):
    """Differential test over option combinations nobody wrote down: 48 seeded draws from the whole option space (sampler, film,
    MIS, bounce mode, tri-light mode, NEE weights, clamping, regularisation, roulette, hidden lights, white first-bounce albedo,
    transparent background, samples per pixel and per pass, depth of field, pre-transformed vertices, seed) on the zoo scene."""
    sc = _zoo_scene()
    ss = R.SceneStage(ctx, sc)
    osc = oracle.OracleScene(sc)
    # TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS: longer campaigns with other seeds (run by hand; profiles/r3/fuzz_campaign.txt)
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "2024")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS", "48"))):
        per_pass = int(rng.choice([1, 1, 2, 3]))
        kw = dict(
            max_bounces=int(rng.integers(1, 7)), sampler=int(rng.integers(0, 4)), film=int(rng.integers(0, 3)), film_radius=float(rng.choice([0.5, 1.0, 1.5])),
            mis_mode=int(rng.integers(0, 3)), bounce_mode=int(rng.integers(0, 3)), tri_light_mode=int(rng.integers(0, 3)),
            nee_point=float(rng.choice([0.0, 1.0, 2.5])), nee_directional=float(rng.choice([0.0, 1.0, 0.3])), nee_envmap=float(rng.choice([0.0, 1.0])),
            indirect_clamping=float(rng.choice([0.0, 0.0, 5.0])), regularization_gamma=float(rng.choice([0.0, 0.0, 0.3])),
            russian_roulette_delta=float(rng.choice([0.0, 0.0, 1.5])), hide_lights=int(rng.integers(0, 2)),
            use_white_albedo_on_first_bounce=int(rng.integers(0, 2)), transparent_background=int(rng.integers(0, 2)),
            samples_per_pass=per_pass, samples_per_pixel=per_pass * int(rng.integers(1, 3)), depth_of_field=int(rng.integers(0, 2)),
            pre_transformed_vertices=int(rng.integers(0, 2)), rng_seed=int(rng.choice([0, 0, 77])))
        frames = int(rng.integers(1, 3))
        # every sixth draw through a shading program compiled for it on the spot (hipRTC, 2-3 s each); the others through the general
        # kernels, which take any option set as data - the same bits (test_specialization.py), without the compiler in the loop.
        # Both in the default arithmetic; the strict comparison at IEEE fp32 follows below.
        spec = None if k % 6 == 0 else False
        img = _render_hip(R, ctx, ss, sc, (96, 96), frames=frames, specialize=spec, **kw)
        opt = oracle.options_for_scene(sc, **kw)
        ref = None
        for f in range(frames):
            ref = osc.render_pt(opt, 96, 96, frame_counter=f, samples_accumulated=f * kw["samples_per_pixel"], color=ref)
        nf = ~np.isfinite(ref).all(-1)
        if nf.any():
            # the reference's own NaN (material_bsdf_sample's inf / inf at grazing refraction): rare - no draw of the
            # suite's seed, a handful in a thousand (tools/fuzz_campaign.sh) - and the HIP path must produce it in the same pixels
            # when it computes in the oracle's arithmetic; the other pixels are compared as usual
            assert nf.mean() < 1e-3, f"draw {k}: the oracle produced {int(nf.sum())} non-finite pixels with {kw}"
        strict = _render_hip(R, ctx, ss, sc, (96, 96), frames=frames, ieee=True, specialize=False, **kw)
        if nf.any():
            assert np.array_equal(~np.isfinite(strict).all(-1), nf), f"draw {k}: non-finite pixels differ from the oracle's with {kw}"
            img = np.where(nf[..., None], np.float32(0), img)
            strict = np.where(nf[..., None], np.float32(0), strict)
            ref = np.where(nf[..., None], np.float32(0), ref)
        _compare(strict, ref, f"draw {k}: {kw}, {frames} frame(s) [IEEE shading]", strict=True)
        # the default arithmetic moves a bounce direction by an ulp, and a ray that grazes an edge then hits the neighbouring triangle: at
        # 1-6 samples per pixel such a path is its pixel.  Typically 0-0.1 % of the pixels, 0.25 % in one draw of 900
        # (profiles/r4/fuzz_campaign.txt: Sobol-Z2, depth of field, three samples per pass) - twice the bound of the strict comparison
        _compare(img, ref, f"draw {k}: {kw}, {frames} frame(s) [default shading arithmetic]", max_bad=2 * MAX_BAD_FRACTION)


@pytest.mark.gpu
def test_sixteen_bit_textures_are_sampled_at_sixteen_bits(R, ctx, oracle):
    """A 16-bit PNG stays RGBA16 (the reference's R16G16B16A16Unorm, src/gltf.cc:548-556; csrc/texture.h fetch_rgba16): a horizontal ramp
    of 4096 distinct 16-bit levels over a quad comes back through the albedo feature with (nearly) as many distinct values - an RGBA8 store
    would leave 256 -, every texel centre reads v / 65535 exactly, and the frame equals the oracle's."""
    from tauray_amd import scene as S
    W = 4096
    ramp = np.zeros((2, W, 4), dtype=np.uint16)
    ramp[..., 0] = (np.arange(W, dtype=np.uint32) * 16 + 7)[None, :]
    ramp[..., 1] = 65535 - ramp[..., 0]
    ramp[..., 2] = 12345
    ramp[..., 3] = 65535
    quad = np.zeros(4, dtype=S.VERTEX)
    quad["pos"] = [(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)]
    quad["normal"] = (0, 0, 1)
    quad["tangent"] = (1, 0, 0, 1)
    quad["uv"] = [(0, 1), (1, 1), (1, 0), (0, 0)]
    cam = S.Camera(projection=S.PROJ_ORTHOGRAPHIC)
    cam.ortho = (-1, 1, -1, 1, -10, 10)
    cam.transform = S.trs_matrix((0, 0, 2))
    sc = S.SceneDesc(instances=S.make_instance(np.eye(4), S.make_material(albedo=(1, 1, 1, 1), metallic=0.0, roughness=1.0, albedo_tex=0)),
                     spans=np.array([(0, 4, 0, 2)], dtype=S.MESH_SPAN), vertices=quad, indices=np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32),
                     textures=[ramp], cameras=[cam], point_lights=S.make_point_light((5, 5, 5), (0, 0, 3), 0.0)).finalize(True)
    infos, texels = sc.texture_table()
    assert infos["format"][0] == 1 and len(texels) == 2 * W * 8
    ss = R.SceneStage(ctx, sc)
    # the path tracer's albedo target (feature_stage traces every ray from the camera origin, rt_feature.rgen:35: no use with an
    # orthographic camera), one pixel per texel
    kw = dict(max_bounces=2, projection=S.PROJ_ORTHOGRAPHIC)
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(sc, **kw), _dup((W, 4)))
    bufs = {n: ctx.alloc(W * 4 * 16).zero() for n in ("color", "albedo")}
    pt.run_targets(bufs)
    img = bufs["albedo"].download((4, W, 4))
    pt.close()
    ref = oracle.OracleScene(sc).render_pt_targets(oracle.options_for_scene(sc, **kw), W, 4, ["color", "albedo"])["albedo"][0]
    assert np.abs(img - ref).max() < 1e-6
    # texel centres: pixel x looks at texel x; sRGB decode (inverse_srgb_correction) applies to the colour channels
    lin = lambda c: np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    want = lin(ramp[0, :, 0].astype(np.float64) / 65535.0)
    assert np.abs(img[1, :, 0] - want).max() < 2e-6
    assert len(np.unique(img[1, :, 0])) > 4000      # an RGBA8 store would leave 256 levels
    _compare(_render_hip(R, ctx, ss, sc, (256, 64), max_bounces=2, projection=S.PROJ_ORTHOGRAPHIC),
             oracle.OracleScene(sc).render_pt(oracle.options_for_scene(sc, max_bounces=2, projection=S.PROJ_ORTHOGRAPHIC), 256, 64), "16-bit albedo texture")


@pytest.mark.gpu
def test_scenes_without_triangle_records_render_the_same(R, ctx):
    """The command-line k_shade reads a hit's vertices from per-triangle records addressed by index_offset / 3 + primitive (csrc/common.h
    ShadeTri).  A scene whose spans do not start at whole triangles, or whose meshes share indices over different vertices, gets no
    records and the general kernel: the same frame bit for bit at IEEE fp32 (the general kernel's arithmetic)."""
    import copy
    from tauray_amd.gltf import load_glb
    base = load_glb(os.path.join(GOLDEN, "test.glb"), 96, 96)
    shifted = copy.copy(base)      # one unused index in front: every span starts one index later
    shifted.indices = np.concatenate([np.zeros(1, np.uint32), base.indices])
    shifted.spans = base.spans.copy()
    shifted.spans["index_offset"] += 1
    # a second, different copy of the vertices (other normals and texture coordinates) used by the odd instances - over the same indices
    # (no records possible: one index triangle, two sets of vertices) and over a copy of the indices (records possible)
    other = base.vertices.copy()
    other["normal"] = other["normal"][:, [1, 2, 0]]
    other["uv"] = 1.0 - other["uv"]
    twin = copy.copy(base)
    twin.vertices = np.concatenate([base.vertices, other])
    twin.spans = base.spans.copy()
    twin.spans["vertex_offset"][1::2] += len(base.vertices)
    twin_own = copy.copy(twin)
    twin_own.indices = np.concatenate([base.indices, base.indices])
    twin_own.spans = twin.spans.copy()
    twin_own.spans["index_offset"][1::2] += len(base.indices)
    frames = {}
    for tag, sc in (("records", base), ("shifted indices", shifted), ("shared indices", twin), ("own indices", twin_own)):
        ss = R.SceneStage(ctx, sc)
        frames[tag] = _render_hip(R, ctx, ss, sc, (96, 96), ieee=True, max_bounces=3)
        assert np.isfinite(frames[tag]).all() and frames[tag][..., :3].mean() > 1e-3
    for a, b in (("shifted indices", "records"), ("shared indices", "own indices")):
        assert np.array_equal(frames[a], frames[b]), f"{a} vs {b}: {int((frames[a] != frames[b]).any(-1).sum())} pixels differ"
    assert not np.array_equal(frames["own indices"], frames["records"]), "the second set of vertices changed nothing"


@pytest.mark.gpu
def test_triangle_counts_around_the_one_workgroup_clustering(R, ctx, oracle, monkeypatch):
    """The builder clusters the last 1 024 clusters in one workgroup (csrc/bvh_build.hip k_ploc_tail) - scenes of 1 024 triangles or fewer
    never see a grid round.  Hit parity (bit-exact closest hits, equal visibility) at the sizes where that path starts, ends and is skipped,
    with the one-workgroup rounds and with every round as grid launches (TRHIP_PLOC_NO_TAIL=1)."""
    from tauray_amd import scene as S
    for n in (1, 2, 3, 5, 63, 64, 65, 1000, 1023, 1024, 1025, 1500, 2047, 2049, 5000):
        rng = np.random.default_rng(1000 + n)
        centre = rng.normal(size=(n, 3)) * 3.0
        tri = (centre[:, None, :] + rng.normal(size=(n, 3, 3)) * 10.0 ** rng.uniform(-2, 0.5, size=(n, 1, 1))).astype(np.float32)
        verts = np.zeros(3 * n, dtype=S.VERTEX)
        verts["pos"] = tri.reshape(-1, 3)
        verts["normal"] = (0, 0, 1)
        verts["tangent"] = (1, 0, 0, 1)
        cam = S.Camera(fov=60, aspect=1.0)
        cam.transform = S.trs_matrix((0, 0, 30))
        sc = S.SceneDesc(instances=S.make_instance(np.eye(4), S.make_material(albedo=(0.5, 0.5, 0.5, 1.0), metallic=0.0, roughness=0.5, double_sided=True)),
                         spans=np.array([(0, 3 * n, 0, n)], dtype=S.MESH_SPAN), vertices=verts, indices=np.arange(3 * n, dtype=np.uint32), cameras=[cam]).finalize(True)
        osc = oracle.OracleScene(sc)
        m = 8000
        org = (rng.normal(size=(m, 3)) * 6.0).astype(np.float32)
        d = (-org + rng.normal(size=(m, 3)) * 3.0).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([org, np.full((m, 1), 1e-4, np.float32), d, np.full((m, 1), np.inf, np.float32)], axis=1).astype(np.float32)
        srays = rays.copy()
        srays[:, 7] = rng.uniform(0.5, 20.0, size=m)
        o, os_ = osc.trace_closest(rays, None), osc.trace_shadow(srays)
        for mode in ("tail", "grid"):
            if mode == "grid": monkeypatch.setenv("TRHIP_PLOC_NO_TAIL", "1")
            ss = R.SceneStage(ctx, sc)
            monkeypatch.delenv("TRHIP_PLOC_NO_TAIL", raising=False)
            assert ss.accel["leaf_count"] == n and ss.accel["triangle_count"] == n
            g = ss.trace_closest(rays, None)
            for k in ("instance_id", "primitive_id"):
                assert np.array_equal(g[k], o[k]), f"{n} triangles, {mode}: {k}"
            assert np.array_equal(g["t"].view(np.uint32), o["t"].view(np.uint32)), f"{n} triangles, {mode}: t"
            gs = ss.trace_shadow(srays)
            assert np.array_equal(gs == 0, os_ == 0), f"{n} triangles, {mode}: visibility"
        if n >= 5: assert (o["instance_id"] >= 0).mean() > 0.02, f"{n} triangles: the rays miss the cloud"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 214, 215] + [int(x) for x in os.environ.get("TRHIP_FUZZ_SOUPS", "").split()])
def test_random_triangle_soups(R, ctx, oracle, seed, monkeypatch):
    """Hit parity on geometry no modeller would export: 20 000 random triangles of wildly different sizes (1e-3 .. 1e2), needles,
    zero-area triangles, exact duplicates, coplanar overlapping sheets, a few non-opaque instances; closest-hit and shadow
    queries from inside and outside the cloud, with and without stochastic alpha."""
    from tauray_amd import scene as S
    rng = np.random.default_rng(seed)
    n = 20_000
    centre = rng.normal(size=(n, 3)) * rng.choice([0.5, 3.0, 30.0], size=(n, 1))
    size = 10.0 ** rng.uniform(-3, 2, size=(n, 1, 1))
    tri = centre[:, None, :] + rng.normal(size=(n, 3, 3)) * size
    needles = rng.choice(n, 500, replace=False)
    tri[needles, 2] = tri[needles, 1] + (tri[needles, 1] - tri[needles, 0]) * 1e-4 + rng.normal(size=(500, 3)) * 1e-6
    degenerate = rng.choice(n, 300, replace=False)
    tri[degenerate, 2] = tri[degenerate, 0]                                  # zero area
    dup = rng.choice(n, 400, replace=False)
    tri[dup] = tri[(dup + 1) % n]                                            # exact duplicates of other triangles
    # coplanar, overlapping in the plane z = 0.25.  Kept below a few units across: the distance the triangle test computes
    # for a 100-unit triangle hit from 3 mm away is good to ~4e-4 only, far more than the slab test's pad, and which of such
    # sheets wins then depends on the traversal order in both implementations
    sheet = rng.choice(np.where(size[:, 0, 0] < 1.0)[0], 600, replace=False)
    tri[sheet, :, 2] = 0.25
    tri = tri.astype(np.float32)
    verts = np.zeros(3 * n, dtype=S.VERTEX)
    verts["pos"] = tri.reshape(-1, 3)
    verts["normal"] = (0, 0, 1)
    verts["tangent"] = (1, 0, 0, 1)
    parts = 8
    per = n // parts
    insts, spans = [], []
    for k in range(parts):
        alpha = 0.4 if k in (2, 5) else 1.0
        insts.append(S.make_instance(np.eye(4), S.make_material(albedo=(0.5, 0.5, 0.5, alpha), metallic=0.0, roughness=0.5, double_sided=True)))
        spans.append((3 * per * k, 3 * per, 3 * per * k, per))
    cam = S.Camera(fov=60, aspect=1.0)
    cam.transform = S.trs_matrix((0, 0, 100))
    sc = S.SceneDesc(instances=np.concatenate(insts), spans=np.array(spans, dtype=S.MESH_SPAN), vertices=verts,
                     indices=np.tile(np.arange(3 * per, dtype=np.uint32), parts), cameras=[cam]).finalize(True)
    ss = R.SceneStage(ctx, sc)
    assert ss.accel["leaf_count"] == n
    osc = oracle.OracleScene(sc)
    m = 60_000
    org = np.concatenate([rng.normal(size=(m // 2, 3)) * 2.0, rng.normal(size=(m // 2, 3)) * 60.0]).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d[m // 2:] = -org[m // 2:] + rng.normal(size=(m // 2, 3)).astype(np.float32) * 5      # outside rays aim at the cloud
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((m, 1), 1e-4, np.float32), d, np.full((m, 1), np.inf, np.float32)], axis=1).astype(np.float32)
    seeds = rng.integers(0, 2**32, size=m, dtype=np.uint64).astype(np.uint32)
    for sd in (seeds, None):
        g, o = ss.trace_closest(rays, sd), osc.trace_closest(rays, sd)
        same = (g["instance_id"] == o["instance_id"]) & (g["primitive_id"] == o["primitive_id"]) & (g["t"].view(np.uint32) == o["t"].view(np.uint32))
        # The only disagreement allowed: a different member of the coplanar sheet at (nearly) the same distance.  The triangle
        # test's distance is good to ~1e-4 for rays that start a millimetre from a sheet, so which of the overlapping
        # triangles reports the smallest one depends on the order the trees offer them in; everything else must be equal.
        sheet_set = np.zeros(n, dtype=bool); sheet_set[sheet] = True
        for i in np.where(~same)[0]:
            kg, ko = int(g["instance_id"][i]) * per + int(g["primitive_id"][i]), int(o["instance_id"][i]) * per + int(o["primitive_id"][i])
            assert g["instance_id"][i] >= 0 and o["instance_id"][i] >= 0 and sheet_set[kg] and sheet_set[ko], f"ray {i}: {g[i]} vs {o[i]}"
            # how far apart the two distances may be: the triangle test subtracts the ray origin from vertices up to ~100 units
            # away, so each distance carries a few ulps *of those coordinates* - 3e-6 for a ray 1.5 mm from a sheet triangle at
            # 13 units (seed 214, ray 7593: 0.0015588814 against 0.0015557635), which is 2e-3 of the distance itself
            coord = max(float(np.abs(tri[kg]).max()), float(np.abs(tri[ko]).max()), float(np.abs(rays[i, :3]).max()))
            assert abs(float(g["t"][i]) - float(o["t"][i])) <= max(1e-3 * float(o["t"][i]), 8 * 2.0 ** -23 * coord), f"ray {i}: {g[i]} vs {o[i]} (largest coordinate {coord})"
        # how many: rays that start within such a distance of the sheet - 0 to 7 of 60 000 over the 19 seeds run so far (seed 215: 7)
        assert (~same).sum() <= m // 5_000, f"{int((~same).sum())} of {m} closest hits differ"
        assert (g["instance_id"] >= 0).mean() > 0.3
    srays = rays.copy()
    srays[:, 7] = rng.uniform(0.5, 80.0, size=m)
    gs, os_ = ss.trace_shadow(srays), osc.trace_shadow(srays)
    assert np.array_equal(gs == 0, os_ == 0) and np.allclose(gs, os_, atol=1e-6)
    assert ss.accel["triangle_count"] == n
    if seed == 1:      # and whole paths through it: lit, finite, equal to the oracle within the usual tolerance
        import copy
        lit = copy.copy(sc)
        lit.point_lights = S.make_point_light((3e6, 3e6, 3e6), (100, 300, 400), 5.0)      # outside the cloud, like the camera
        lit.envmap = np.ones((2, 4, 4), dtype=np.float32)
        lit.environment_factor = (0.2, 0.25, 0.3, 1.0)
        lit.cameras = [copy.deepcopy(cam)]
        lit.cameras[0].fov = 35
        lit.cameras[0].transform = S.trs_matrix((5, 10, 420))
        ss = R.SceneStage(ctx, lit)
        osc = oracle.OracleScene(lit)
        kw = dict(max_bounces=4, samples_per_pixel=2)
        ref = osc.render_pt(oracle.options_for_scene(lit, **kw), 128, 128)
        assert np.isfinite(ref).all() and ref[..., :3].mean() > 1e-3
        # needles and edge-on giants make shading ill-conditioned for half a percent of the paths (normals from nearly collinear
        # edges): the usual bound at IEEE fp32, 1.5 % of such pixels with the default shading arithmetic (csrc/shade_fast.hip)
        _compare(_render_hip(R, ctx, ss, lit, (128, 128), ieee=True, **kw), ref, "paths through the soup, IEEE shading")
        _compare(_render_hip(R, ctx, ss, lit, (128, 128), **kw), ref, "paths through the soup, default shading arithmetic", max_bad=0.015)


@pytest.mark.gpu
def test_refit_sequences_equal_rebuilds(R, ctx):
    """Twelve random instance-transform updates in a row with skinning steps in between, once with nothing but refits (the
    tree of the first build gets staler every step) and once rebuilding at every step: the same frames, bit for bit."""
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene import from_glm, to_glm, trs_matrix

    seed, steps = int(os.environ.get("TRHIP_FUZZ_SEED", "12")), int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "12"))

    def sequence(refit):
        rng = np.random.default_rng(seed)
        scene = load_glb(os.path.join(GOLDEN, "test.glb"), 96, 96)
        ss = R.SceneStage(ctx, scene)
        sp = scene.spans[4]
        bind = scene.vertices[sp["vertex_offset"]:sp["vertex_offset"] + sp["vertex_count"]].copy()
        skins, _ = _bend_rig(bind, 0.0)
        ss.set_skin(4, skins)
        base = [from_glm(scene.instances["model"][i]) for i in range(len(scene.instances))]
        frames = []
        for step in range(steps):
            for i in rng.choice(np.arange(4, len(scene.instances)), size=2, replace=False):
                m = trs_matrix(rng.uniform(-0.4, 0.4, 3), rng.normal(size=4) * (0.2, 0.2, 0.2, 1.0) + (0, 0, 0, 1), rng.uniform(0.7, 1.3, 3)) @ base[i]
                scene.instances["model_prev"][i] = scene.instances["model"][i]
                scene.instances["model"][i] = to_glm(m)
                scene.instances["model_normal"][i] = to_glm(np.linalg.inv(m).T)
            ss.update_instances(scene.instances, refit=refit)
            if step % 3 == 1:
                ss.skin(4, _bend_rig(bind, float(rng.uniform(-1, 1)))[1], refit=refit)
            frames.append(_render_hip(R, ctx, ss, scene, (96, 96), max_bounces=3))
        return frames

    a, b = sequence(True), sequence(False)
    for step, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), f"step {step}"
    assert not np.array_equal(a[0], a[-1])


@pytest.mark.gpu
def test_random_cameras(R, ctx, oracle):
    """Twenty seeded cameras on the zoo scene: perspective with pan (fov offsets) and depth of field (circular and polygonal
    apertures), orthographic windows off the axis, equirectangular with partial fields of view, rolled and tilted transforms,
    wide and narrow aspect ratios: ray generation and the whole frame against the oracle."""
    import copy
    from tauray_amd import scene as S
    base = _zoo_scene()
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "31")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "20"))):
        sc = copy.copy(base)
        cam = S.Camera()
        kind = k % 3
        w, h = [(96, 96), (160, 64), (48, 120)][int(rng.integers(0, 3))]
        dof = 0
        if kind == 0:
            cam.projection = S.PROJ_PERSPECTIVE
            cam.fov = float(rng.uniform(20, 110))
            cam.fov_offset = (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)))
            if k % 2 == 0:
                cam.set_focus(float(rng.uniform(0.5, 4)), float(rng.uniform(2, 6)), int(rng.choice([0, 3, 6])), float(rng.uniform(0, 90)), 0.036)
                dof = 1
        elif kind == 1:
            cam.projection = S.PROJ_ORTHOGRAPHIC
            cx, cy, hw = rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(1.0, 2.5)
            cam.ortho = (cx - hw, cx + hw, cy - hw, cy + hw, 0.01, 50.0)
        else:
            cam.projection = S.PROJ_EQUIRECTANGULAR
            cam.equirect_fov = (float(rng.uniform(90, 360)), float(rng.uniform(60, 180)))
        cam.transform = S.trs_matrix(rng.uniform(-0.5, 0.5, 3) + (0, 0, 4.0), rng.normal(size=4) * (0.15, 0.15, 0.3, 1.0) + (0, 0, 0, 1.5))
        cam.set_aspect(w / h)
        sc.cameras = [cam]
        ss = R.SceneStage(ctx, sc)
        osc = oracle.OracleScene(sc)
        kw = dict(max_bounces=3, depth_of_field=dof, projection=cam.projection, film=int(rng.integers(0, 3)))
        img = _render_hip(R, ctx, ss, sc, (w, h), **kw)
        ref = osc.render_pt(oracle.options_for_scene(sc, **kw), w, h)
        assert np.isfinite(ref).all(), f"camera {k}"
        _compare(img, ref, f"camera {k} (kind {kind}, {w}x{h}, dof {dof})")
        fs = R.FeatureStage(ctx, ss, 5, _dup((w, h)), projection=cam.projection)
        buf = ctx.alloc(w * h * 16).zero()
        fs.run(buf)
        g, r = buf.download((h, w, 4)), osc.render_feature(5, w, h, projection=cam.projection)
        if kind != 2:
            assert np.array_equal(g, r, equal_nan=True), f"camera {k}: primary hit distances"
        else:       # equirectangular rays go through sin / cos, which differ by ulps between the two maths libraries
            both = np.isfinite(g[..., 0]) & np.isfinite(r[..., 0])
            assert (np.isfinite(g[..., 0]) != np.isfinite(r[..., 0])).mean() <= 2e-3
            assert (np.abs(g[both] - r[both]) > 1e-4 * np.abs(r[both]) + 1e-5).mean() <= 2e-3, f"camera {k}: primary hit distances"


@pytest.mark.gpu
def test_random_lights(R, ctx, oracle):
    """Sixteen seeded light rigs on the zoo scene: up to five point / spot lights (radius 0 .. 0.6, cones of 3 .. 80 degrees,
    falloff exponents 0 .. 8, some inside geometry or behind the camera), up to three directional lights (angles 0 .. 25
    degrees), with and without the environment map, tri-light modes and MIS modes drawn along."""
    import copy
    from tauray_amd import scene as S
    base = _zoo_scene()
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "77")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "16"))):
        sc = copy.copy(base)
        pls = []
        for _ in range(int(rng.integers(0, 6))):
            col = tuple(rng.uniform(2, 60, 3))
            pos = tuple(rng.uniform(-2.5, 2.5, 2)) + (float(rng.uniform(-1.0, 5.0)),)
            radius = float(rng.choice([0.0, 0.0, rng.uniform(0.02, 0.6)]))
            if rng.uniform() < 0.5:
                pls.append(S.make_point_light(col, pos, radius))
            else:
                pls.append(S.make_spotlight(col, pos, tuple(rng.normal(size=3)), radius, float(rng.uniform(3, 80)), float(rng.uniform(0, 8))))
        dls = [S.make_directional_light(tuple(rng.uniform(0.2, 3, 3)), tuple(rng.normal(size=3) * (1, 1, 0.3) + (0, 0, -1)), float(rng.choice([0.0, rng.uniform(0.1, 25)])))
               for _ in range(int(rng.integers(0, 4)))]
        sc.point_lights = np.concatenate(pls) if pls else np.zeros(0, dtype=S.POINT_LIGHT)
        sc.directional_lights = np.concatenate(dls) if dls else np.zeros(0, dtype=S.DIRECTIONAL_LIGHT)
        if k % 3 == 0:
            sc.envmap, sc.environment_factor = None, (0, 0, 0, 0)
        ss = R.SceneStage(ctx, sc)
        osc = oracle.OracleScene(sc)
        kw = dict(max_bounces=int(rng.integers(2, 5)), mis_mode=int(rng.integers(0, 3)), tri_light_mode=int(rng.integers(0, 3)), hide_lights=int(rng.integers(0, 2)),
                  sampler=int(rng.integers(0, 2)))
        img = _render_hip(R, ctx, ss, sc, (112, 112), **kw)
        ref = osc.render_pt(oracle.options_for_scene(sc, **kw), 112, 112)
        assert np.isfinite(ref).all(), f"rig {k}: the oracle produced a non-finite pixel"
        _compare(img, ref, f"rig {k}: {len(pls)} point / spot, {len(dls)} directional, {kw}")


@pytest.mark.gpu
def test_random_materials(R, ctx, oracle):
    """Twelve seeded draws of all nine panel materials of the zoo scene: metallic and roughness anywhere in [0, 1] (with
    exact zeros and ones), transmittance, ior in [0.5, 3] away from 1, albedo alpha, emission (emissive panels become tri
    lights), single- and double-sided, normal factors - whole frames and the material gbuffer targets against the oracle."""
    import copy
    from tauray_amd import scene as S
    base = _zoo_scene()
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "5150")))
    corner = lambda: float(rng.choice([0.0, 1.0, rng.uniform(0, 1), rng.uniform(0, 1)]))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "12"))):
        sc = copy.copy(base)
        sc.instances = base.instances.copy()
        for i in range(9):
            ior = float(rng.uniform(0.5, 3.0))
            if abs(ior - 1.0) < 0.03:
                ior = 1.3
            emis = tuple(rng.uniform(0, 4, 3)) if rng.uniform() < 0.25 else (0, 0, 0)
            sc.instances["mat"][i] = S.make_material(albedo=tuple(rng.uniform(0, 1, 3)) + (float(rng.choice([1.0, 1.0, rng.uniform(0.1, 0.9)])),),
                                                     metallic=corner(), roughness=corner(), emission=emis,
                                                     transmittance=float(rng.choice([0.0, 0.0, 1.0, rng.uniform(0, 1)])), ior=ior,
                                                     normal_factor=float(rng.uniform(0.5, 1.5)), double_sided=bool(rng.integers(0, 2)))
        sc.finalize(True)
        ss = R.SceneStage(ctx, sc)
        osc = oracle.OracleScene(sc)
        assert np.array_equal(ss.tri_lights().view(np.uint8), osc.tri_lights().view(np.uint8))
        kw = dict(max_bounces=int(rng.integers(2, 6)), sampler=int(rng.integers(0, 2)), tri_light_mode=int(rng.integers(0, 3)))
        img = _render_hip(R, ctx, ss, sc, (112, 112), **kw)
        ref = osc.render_pt(oracle.options_for_scene(sc, **kw), 112, 112)
        assert np.isfinite(ref).all(), f"draw {k}: the oracle produced a non-finite pixel"
        _compare(img, ref, f"draw {k}: {kw}")
        got = _render_targets_hip(R, ctx, ss, sc, (112, 112), ["material", "albedo"], max_bounces=2)
        want = osc.render_pt_targets(oracle.options_for_scene(sc, max_bounces=2), 112, 112, ["material", "albedo"])
        assert np.allclose(got["material"], want["material"], atol=1e-6) and np.allclose(got["albedo"], want["albedo"], atol=1e-6), f"draw {k}"


@pytest.mark.gpu
def test_random_direct_and_gbuffer_targets(R, ctx, oracle):
    """Seeded draws over the two stages that write a gbuffer - direct_stage and path_tracer_stage with targets - on the zoo scene:
    samples per pass and per pixel, sampler, film filter, tri-light mode, NEE weights, hidden lights, white first-bounce albedo,
    transparent background, bounces, one or two accumulated frames; colour, the demodulated diffuse / reflection targets and the
    first-hit AOVs against the oracle.  TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS_SMALL run longer campaigns (tools/fuzz_campaign.sh)."""
    sc = _zoo_scene()
    ss = R.SceneStage(ctx, sc)
    osc = oracle.OracleScene(sc)
    names = ["color", "diffuse", "reflection", "albedo", "normal", "pos", "instance_id"]
    W = H = 96
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "23")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "8"))):
        direct = bool(rng.integers(0, 2))
        per_pass = int(rng.choice([1, 1, 2, 4]))
        kw = dict(max_bounces=int(rng.integers(1, 5)), sampler=int(rng.integers(0, 4)), film=int(rng.integers(0, 3)), film_radius=float(rng.choice([0.5, 1.0])),
                  tri_light_mode=int(rng.integers(0, 3)), nee_point=float(rng.choice([0.0, 1.0, 2.5])), nee_directional=float(rng.choice([0.0, 1.0])),
                  hide_lights=int(rng.integers(0, 2)), use_white_albedo_on_first_bounce=int(rng.integers(0, 2)), transparent_background=int(rng.integers(0, 2)),
                  samples_per_pass=per_pass, samples_per_pixel=per_pass * int(rng.integers(1, 3)), rng_seed=int(rng.choice([0, 0, 5])))
        frames = int(rng.integers(1, 3))
        # a third of the draws with the shading kernels at IEEE arithmetic.  The AOVs of a jittered first hit (film filters draw the
        # sub-pixel offset with sin / cos, ocml's against glibc's in the oracle) agree to a few 1e-5 then, to 1e-4 with the default arithmetic
        ieee = k % 3 == 0
        what = f"draw {k}: {'direct' if direct else 'path tracer'} {kw}, {frames} frame(s), {'IEEE' if ieee else 'default'} arithmetic"
        st = (R.DirectStage if direct else R.PathTracerStage)(ctx, ss, R.options_for_scene(sc, **kw), _dup((W, H)))
        st.set_shading_arithmetic(ieee)
        bufs = {n: ctx.alloc(W * H * R.PathTracerStage.TARGETS[n][0] * 4).zero() for n in names}
        for _ in range(frames):
            st.run_targets(bufs)
        got = {n: np.frombuffer(bufs[n].download((1, H, W, R.PathTracerStage.TARGETS[n][0])).tobytes(), dtype=R.PathTracerStage.TARGETS[n][1])
                  .reshape(1, H, W, R.PathTracerStage.TARGETS[n][0]) for n in names}
        assert st.counters()["stack_overflows"] == 0
        st.close()
        oopt = oracle.options_for_scene(sc, **kw)
        ref = None
        for f in range(frames):
            ref = osc.render_pt_targets(oopt, W, H, names, frame_counter=f, samples_accumulated=kw["samples_per_pixel"] * f, targets=ref, direct=direct)
        nf = ~np.isfinite(ref["color"]).all(-1)
        if nf.any():      # the reference's own NaN: rare; those pixels are left out here (test_random_option_combinations pins them)
            assert nf.mean() < 1e-3, what
            for n in ("color", "diffuse", "reflection"):
                got[n] = np.where(nf[..., None], np.float32(0), got[n]); ref[n] = np.where(nf[..., None], np.float32(0), ref[n])
        _compare(got["color"], ref["color"], what + ": color")
        assert np.array_equal(got["instance_id"], ref["instance_id"]), what
        for n, tol in (("albedo", 1e-6), ("normal", 5e-5 if ieee else 1e-4), ("pos", 5e-5 if ieee else 1e-4)):
            assert float(np.nanmax(np.abs(got[n] - ref[n]))) <= tol * max(1.0, float(np.nanmax(np.abs(ref[n])))), f"{what}: {n}"
        for n in ("diffuse", "reflection"):
            rel = np.abs(got[n][..., :3] - ref[n][..., :3]) / (np.abs(ref[n][..., :3]) + 1e-2)
            assert float((rel.max(-1) > REL_TOL).mean()) <= MAX_BAD_FRACTION, f"{what}: {n}"
            assert float((np.abs(got[n][..., 3] - ref[n][..., 3]) > 1e-4 * (np.abs(ref[n][..., 3]) + 1.0)).mean()) <= MAX_BAD_FRACTION, f"{what}: {n} alpha"


# ----------------------------------------------------------------------------------------------------------
# Several ranks of one process exchanging partial frames with nothing but stream order between them


def _in_process_job(R, scene, opt, size, world, strategy, F, frames, workloads=None, break_waits=False, B=1):
    """`world` RtRenderer ranks on fake devices (one Context each on HIP device 0) joined by a transfer.LocalExchange: sends
    are device-to-device copies on the default stream, there is no host synchronisation between or inside frames, and
    every frame's display image is copied to a history buffer in stream order.  Returns [frames, H, W, 4]."""
    from tauray_amd import _lib
    from tauray_amd.transfer import LocalExchange
    W, H = size
    ex = LocalExchange(world)
    ctxs = [R.Context(0) for _ in range(world)]
    rrs = [R.RtRenderer(ctxs[r], scene, opt, size, strategy=strategy, rank=r, world_size=world, exchange=ex, frames_in_flight=F, frames_per_launch=B)
           for r in range(world)]
    if workloads is not None:
        for rr in rrs:
            rr.set_device_workloads(workloads)
    if break_waits:      # negative control: the non-display ranks' sends no longer wait for their path tracing
        for c in ctxs[1:]:
            c.stream_wait = lambda stream, on: None
    hist = ctxs[0].alloc(frames * W * H * 16).zero()
    ctxs[0].sync()
    for f in range(0, frames, B):      # B frames per render(): B layers of the display image
        for r in list(range(1, world)) + [0]:
            rrs[r].render()
        slot_stream = rrs[0].current.stream if world == 1 else None
        if slot_stream is not None:      # a single rank keeps the whole frame on its slot's stream (RtRenderer.render): order the copy behind it
            ctxs[0].stream_wait(None, slot_stream)
        rc = _lib.lib().trhip_copy_peer(ctxs[0].h, hist.data_ptr() + f * W * H * 16, ctxs[0].h, rrs[0].display.data_ptr(), B * W * H * 16, None)
        assert rc == 0
        if slot_stream is not None:      # ... and the slot's next frame behind the copy
            ctxs[0].stream_wait(slot_stream, None)
    for rr in rrs:
        rr.sync()
    out = hist.download((frames, H, W, 4))
    for rr in rrs:
        assert rr.counters()["stack_overflows"] == 0
        rr.close()
    return out


def _single_rank_frames(R, ctx, scene, opt, size, frames):
    rr = R.RtRenderer(ctx, scene, opt, size, use_torch=False)
    out = []
    for _ in range(frames):
        rr.render()
        out.append(rr.download("display")[0])
    rr.close()
    return np.stack(out)


@pytest.mark.gpu
@pytest.mark.parametrize("world,strategy,F,B", [(3, 1, 4, 1), (4, 2, 4, 1), (2, 1, 1, 1), (8, 1, 3, 1), (4, 2, 3, 5), (8, 1, 2, 2)])
def test_in_process_ranks_exchange_is_stream_ordered(R, ctx, world, strategy, F, B):
    """The bracket of stream dependencies around the exchange in RtRenderer.render (path tracing on the slot stream ->
    default stream: send / stitch / tonemap -> slot stream) is all that orders 50 consecutive frames with four in flight:
    every stitched, tonemapped frame equals the single-rank frame of the same index bit for bit.  The reference's
    counterpart is the timeline-semaphore chain of src/rt_renderer.cc:84-133."""
    from tauray_amd.gltf import load_glb
    W, H, frames = 160, 96, 50
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    opt = R.options_for_scene(scene, max_bounces=3)
    ref = _single_rank_frames(R, ctx, scene, opt, (W, H), frames)
    assert not np.array_equal(ref[0], ref[1])      # the sample counter advances: frames are distinguishable
    got = _in_process_job(R, scene, opt, (W, H), world, strategy, F, frames, B=B)      # B > 1: frame batches (frames_per_launch)
    wrong = [f for f in range(frames) if not np.array_equal(got[f], ref[f])]
    assert not wrong, f"frames {wrong[:10]} differ from the single-rank frames"


@pytest.mark.gpu
def test_random_in_process_jobs(R, ctx):
    """Seeded draws of multi-rank jobs in one process: 1-8 ranks on fake devices, scanlines or strips, 1-6 frames in flight, 1-4
    frames per launch, even or load-balancer shares (zeros included), frame sizes that are not multiples of anything - nothing but
    stream order between the ranks' path tracing, their sends, the stitch and the tonemap; every display frame must equal the
    single-rank frame of the same index bit for bit.  TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS_SMALL run longer campaigns."""
    from tauray_amd.gltf import load_glb
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "13")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "6"))):
        scale = int(os.environ.get("TRHIP_FUZZ_SCALE", "1"))      # larger frames: longer kernels, more room for a missing dependency to show
        W, H = scale * int(rng.integers(24, 320)), scale * int(rng.integers(16, 200))
        world = int(rng.integers(1, 9))
        strategy = int(rng.choice([1, 2]))
        F, B = int(rng.integers(1, 7)), int(rng.choice([1, 1, 2, 3, 4]))
        frames = B * int(rng.integers(2, 7))
        workloads = None
        if world > 1 and rng.uniform() < 0.5:
            w = rng.uniform(0.05, 1, world) * (rng.uniform(0, 1, world) > 0.15)
            workloads = (w / w.sum()).tolist() if w.sum() > 0 else None
        scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
        opt = R.options_for_scene(scene, max_bounces=int(rng.integers(1, 4)))
        ref = _single_rank_frames(R, ctx, scene, opt, (W, H), frames)
        got = _in_process_job(R, scene, opt, (W, H), world, strategy, F, frames, workloads=workloads, B=B)
        wrong = [f for f in range(frames) if not np.array_equal(got[f], ref[f])]
        assert not wrong, (f"draw {k}: {W}x{H}, {world} ranks, strategy {strategy}, {F} in flight, {B} per launch, shares {workloads}: "
                           f"frames {wrong[:10]} of {frames} differ from the single-rank frames")


@pytest.mark.gpu
def test_in_process_exchange_without_its_waits_is_caught(R, ctx):
    """Negative control for the test above: with the non-display ranks' stream dependencies removed their sends overtake
    their path tracing and ship stale pixels - the comparison notices."""
    from tauray_amd.gltf import load_glb
    W, H, frames = 480, 272, 12
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    opt = R.options_for_scene(scene, max_bounces=4)
    ref = _single_rank_frames(R, ctx, scene, opt, (W, H), frames)
    got = _in_process_job(R, scene, opt, (W, H), 2, 1, 4, frames, break_waits=True)
    assert any(not np.array_equal(got[f], ref[f]) for f in range(frames))


@pytest.mark.gpu
def test_set_device_workloads_resizes_python_shares(R, ctx):
    """RtRenderer.set_device_workloads (src/rt_renderer.cc:135-183): shares that grow past the even split get new targets and
    receive buffers of the new size; the stitched frame is still the single-rank frame (the old code kept the
    images of the even split and wrote past them)."""
    from tauray_amd.gltf import load_glb
    W, H, frames = 96, 64, 4
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    opt = R.options_for_scene(scene, max_bounces=3)
    ref = _single_rank_frames(R, ctx, scene, opt, (W, H), frames)
    for workloads in ([0.1, 0.2, 0.7], [0.05, 0.9, 0.05], [0.5, 0.0, 0.5]):
        got = _in_process_job(R, scene, opt, (W, H), 3, 2, 1, frames, workloads=workloads)
        assert np.array_equal(got, ref), workloads
    with pytest.raises(ValueError):
        R.RtRenderer(ctx, scene, R.options_for_scene(scene, samples_per_pixel=2), (W, H), rank=0, world_size=2, use_torch=False, shard="samples", accumulate=True)


# ----------------------------------------------------------------------------------------------------------
# Every kernel instance and every schedule switch renders the same frame


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["sponza_class_60k", "sponza_teapots"])
def test_profiling_instances_render_the_same_frame(R, ctx, workload):
    """bench.py takes its work counters from the counting instances of the kernels (`k_trace_closest<true, ..>`, `k_shade<true, ..>`)
    and its kernel times from the serialised single-lane schedule (`k_trace_closest<false, true, ..>`, separate shadow launches):
    both must render the frame of the production instances bit for bit, and count the same rays.  In the counting schedule the
    shadow launch of bounce b runs on a side stream next to the closest-hit launch of bounce b + 1: on the million-triangle scene
    traversal stacks reach 26 entries, past the 16 that live in LDS, so the quad tails of both kernels spill - into regions of
    their own (PtStage::render), or this comparison fails."""
    from tauray_amd import scenes
    W, H = 640, 360
    scene = scenes.sponza_class(seed=2, target_tris=60000, width=W, height=H) if workload == "sponza_class_60k" else scenes.sponza_teapots(width=W, height=H)
    ss = R.SceneStage(ctx, scene)
    opt = R.options_for_scene(scene, max_bounces=4)
    frames, counters = {}, {}
    for tag, (count, timing) in {"plain": (False, False), "counting": (True, False), "timed": (False, True), "both": (True, True)}.items():
        pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
        pt.set_profiling(count, timing)
        buf = ctx.alloc(W * H * 16).zero()
        for _ in range(2):
            pt.reset_accumulated_samples()
            pt.run(buf)
        frames[tag] = buf.download((H, W, 4))
        counters[tag] = pt.counters()
        if timing:
            t = pt.timings()
            assert t["trace_closest_launches"] == 8 and t["trace_shadow_launches"] == 6 and t["shade_launches"] == 8 and t["trace_closest_ms"] > 0
        pt.close()
    for tag in ("counting", "timed", "both"):
        assert np.array_equal(frames[tag], frames["plain"]), f"{tag}: {int((frames[tag] != frames['plain']).any(-1).sum())} pixels differ"
        for k in ("closest_rays", "shadow_rays", "stack_overflows"):
            assert counters[tag][k] == counters["plain"][k], (tag, k)
    c = counters["counting"]
    assert c["node_visits"] > 5 * c["closest_rays"] and c["tri_tests"] > c["closest_rays"] and 0 < c["surface_hits"] <= c["closest_rays"]
    # a ray's visit count depends a little on when its wave switches to quads (the two loops order equal-distance children and
    # pending leaves differently), i.e. on which rays share a wave: the multi-lane and the single-lane schedule agree to a few 1e-4
    # (1.2e-4 of the triangle tests at this size on two lanes)
    for k in ("node_visits", "tri_tests"):
        assert abs(counters["both"][k] - c[k]) <= 3e-4 * c[k], k


_SWITCH_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
W, H = 1920, 1080
scene = scenes.sponza_class(seed=2, target_tris=60000, width=W, height=H)
ctx = R.Context(0)
ss = R.SceneStage(ctx, scene)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
buf = ctx.alloc(W * H * 16).zero()
pt.run(buf)
assert pt.counters()["stack_overflows"] == 0
np.save(sys.argv[2], buf.download((H, W, 4)))
"""


@pytest.mark.gpu
def test_schedule_and_experiment_switches_render_the_same_frame(tmp_path):
    """The environment switches of DESIGN.md section 5 select schedules (lanes, fused launches, grids), builders and kernel
    instances (the general k_shade for the last bounce, the general k_shade instead of a specialised one): a full-size frame
    (large enough for the four-lane schedule) is the same bits under every one of them.  (The experiments that used to ride
    here - treetop in LDS, split k_shade, packets, queue reordering, pre-splitting - are patches under experiments/ now.)"""
    import subprocess, sys
    from conftest import ROOT
    script = tmp_path / "render.py"
    script.write_text(_SWITCH_SCRIPT)
    variants = {"default": {}, "one_lane_unfused": {"TRHIP_LANES": "1", "TRHIP_FUSED": "0"}, "no_overlap": {"TRHIP_LANES": "1", "TRHIP_FUSED": "0", "TRHIP_OVERLAP": "0"},
                "two_lanes": {"TRHIP_LANES": "2"}, "small_grids": {"TRHIP_GRID_BLOCKS": "300", "TRHIP_SHADE_BLOCKS": "100"},
                "general_last_bounce": {"TRHIP_SHADE_LAST": "0"},
                "lbvh": {"TRHIP_BUILDER": "lbvh"}, "unoptimised_tree": {"TRHIP_BVH_OPT": "0"}, "greedy_collapse": {"TRHIP_COLLAPSE": "greedy"},
                # no ahead-of-time instance of the command-line set: a program compiled for it (hipRTC / kernel cache), or the general kernels
                "compiled_shade": {"TRHIP_SHADE_CLI": "0"}, "general_shade": {"TRHIP_SHADE_CLI": "0", "TRHIP_SPECIALIZE": "0"},
                "optimised_lbvh": {"TRHIP_BUILDER": "lbvh", "TRHIP_BVH_OPT": "24", "TRHIP_BVH_OPT_MOD": "3"},
                "no_triangle_records": {"TRHIP_NO_SHADE_TRIS": "1"},      # no ShadeTri records: the general k_shade with indexed vertex fetches (IEEE fp32)
                "lanes_enqueued_in_turn": {"TRHIP_ENQUEUE": "step"}, "lanes_enqueued_a_step_apart": {"TRHIP_ENQUEUE": "skew1"},      # instead of lane after lane (the first frame of a stage)
                "ploc_grid_rounds": {"TRHIP_PLOC_NO_TAIL": "1"},      # every clustering round as grid launches (csrc/bvh_build.hip k_ploc_tail otherwise)
                # the shading kernels of the command-line option set exist at IEEE fp32 too (TRHIP_SHADE_FAST=0; csrc/shade_fast.hip)
                "ieee_shade": {"TRHIP_SHADE_FAST": "0"}, "ieee_compiled_shade": {"TRHIP_SHADE_FAST": "0", "TRHIP_SHADE_CLI": "0"},
                "ieee_general_shade": {"TRHIP_SHADE_FAST": "0", "TRHIP_SHADE_CLI": "0", "TRHIP_SPECIALIZE": "0"},
                "ieee_no_triangle_records": {"TRHIP_SHADE_FAST": "0", "TRHIP_NO_SHADE_TRIS": "1"},
                "ieee_general_last_bounce": {"TRHIP_SHADE_FAST": "0", "TRHIP_SHADE_LAST": "0"}}
    # (general_last_bounce is one of them: the last bounce through k_shade<.., LAST = false> happened to render the bits of the LAST
    # instance until a change elsewhere in the parameter block moved the compiler's choices - profiles/r4/lane_balance.txt)
    other_instances = ("compiled_shade", "general_shade", "no_triangle_records", "general_last_bounce")
    # TRHIP_FUZZ_SWITCH_COMBOS=N (with TRHIP_FUZZ_SEED): N random combinations of the switches that keep the default arithmetic, by hand
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "1")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_SWITCH_COMBOS", "0"))):
        pick = lambda key, values: {key: str(rng.choice(values))} if rng.uniform() < 0.5 else {}
        combo = {}
        for key, values in (("TRHIP_LANES", ["1", "2", "3", "4"]), ("TRHIP_FUSED", ["0", "1"]), ("TRHIP_OVERLAP", ["0", "1"]), ("TRHIP_GRID_BLOCKS", ["64", "300", "1024", "4096"]),
                            ("TRHIP_SHADE_BLOCKS", ["32", "100", "2048"]), ("TRHIP_SHADE_LAST", ["0", "1"]), ("TRHIP_BUILDER", ["lbvh", "ploc"]),
                            ("TRHIP_BVH_OPT", ["0", "3", "8"]), ("TRHIP_COLLAPSE", ["greedy", "cost"])):
            combo.update(pick(key, values))
        variants[f"combo{k}_" + "_".join(f"{a[6:]}{b}" for a, b in combo.items())] = combo
    def render(item):
        tag, env = item
        out = str(tmp_path / f"{tag}.npy")
        e = dict(os.environ)
        for k in ("TRHIP_LANES", "TRHIP_FUSED", "TRHIP_OVERLAP", "TRHIP_GRID_BLOCKS", "TRHIP_SHADE_BLOCKS", "TRHIP_SHADE_LAST", "TRHIP_BUILDER", "TRHIP_BVH_OPT", "TRHIP_BVH_OPT_MOD", "TRHIP_COLLAPSE", "TRHIP_SHADE_CLI", "TRHIP_SHADE_FAST", "TRHIP_SPECIALIZE", "TRHIP_PLOC_NO_TAIL", "TRHIP_NO_SHADE_TRIS", "TRHIP_ENQUEUE"):
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([sys.executable, str(script), ROOT, out], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"{tag}: {r.stderr[-1500:]}"
        return tag, np.load(out)

    # a process per variant (the switches are read once per process), four at a time: most of a variant's second is the interpreter
    # and the scene, not the frame
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as pool:
        frames = dict(pool.map(render, variants.items()))
    ref = frames["default"]
    assert np.isfinite(ref).all() and ref[..., :3].mean() > 1e-3
    # two arithmetic modes of k_shade: the default (what Vulkan asks of the reference's GLSL) and IEEE fp32; within a mode every
    # schedule, tree and kernel instance renders the same bits (the general k_shade only exists at IEEE fp32)
    ieee = frames["ieee_shade"]
    for tag, f in frames.items():
        if tag in other_instances:
            # another instance of k_shade at the default arithmetic: an implementation of its own within Vulkan's accuracy (see
            # test_path_tracer_matches_oracle) - close, not the same bits
            _compare(f[None], ref[None], f"{tag} vs the default frame")
            assert (np.abs(f - ref) <= 1e-4 * np.abs(ref) + 1e-6).all(-1).mean() > 0.98, tag
            continue
        base = ieee if tag.startswith("ieee_") else ref
        assert np.array_equal(f, base), f"{tag}: {int((f != base).any(-1).sum())} pixels differ from the {'IEEE' if base is ieee else 'default'} frame"
    _compare(ref[None], ieee[None], "default shading arithmetic vs IEEE fp32")


@pytest.mark.gpu
def test_display_written_by_the_resolve_equals_the_tonemap_stage(R, ctx, monkeypatch):
    """A one-device renderer has nothing between path_tracer_stage and tonemap_stage (src/rt_renderer.cc), and the stage's last pass
    writes the display image while it writes the colour target (trhip_pt_set_fused_tonemap).  Same bits as the tonemap stage run
    afterwards: every operator, the alpha grid over a transparent background, progressive accumulation, several passes per frame,
    frames in flight."""
    from tauray_amd.gltf import load_glb
    W, H = 160, 96
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    cases = [
        (dict(max_bounces=3), dict(), dict()),
        (dict(max_bounces=2, transparent_background=1), dict(tonemap=dict(op=R.TONEMAP_FILMIC, alpha_grid_background=True)), dict()),
        (dict(max_bounces=2), dict(tonemap=dict(op=1, exposure=1.7, gamma=2.4)), dict(accumulate=True)),
        (dict(max_bounces=2, samples_per_pixel=4, samples_per_pass=2), dict(tonemap=dict(op=3)), dict()),
        (dict(max_bounces=2), dict(tonemap=dict(op=4, exposure=0.6)), dict(frames_in_flight=2)),
        (dict(max_bounces=2), dict(tonemap=dict(op=0, exposure=2.0)), dict(frames_in_flight=3, frames_per_launch=2, viewports=1)),
    ]
    for okw, tkw, rkw in cases:
        opt = R.options_for_scene(scene, **okw)
        frames = {}
        for fused in ("0", "1"):
            monkeypatch.setenv("TRHIP_FUSED_TONEMAP", fused)
            rr = R.RtRenderer(ctx, scene, opt, (W, H), use_torch=False, **tkw, **rkw)
            assert rr.fused_tonemap == (fused == "1")
            out = []
            for _ in range(4):
                rr.render()
                if rr.frames_in_flight == 1:
                    out.append((rr.download("color"), rr.download("display")))
            rr.sync()
            if rr.frames_in_flight > 1:
                layers = rr.viewports
                out = [(s.color.download((layers, H, W, 4)), s.display.download((layers, H, W, 4))) for s in rr.slots]
            frames[fused] = out
            rr.close()
        for k, ((c0, d0), (c1, d1)) in enumerate(zip(frames["0"], frames["1"])):
            assert np.array_equal(c0, c1), f"{okw} {tkw} {rkw}: colour {k}"
            assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32)), f"{okw} {tkw} {rkw}: display {k}"
            assert not np.array_equal(d0, c0)
    # the tonemap parameters are the renderer's to edit between frames (tonemap_stage's options in the reference): an edit takes effect on
    # the next frame with the fused write as it does with the stage, and render(tonemap=False) leaves the display image alone
    opt = R.options_for_scene(scene, max_bounces=2)
    shown = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("TRHIP_FUSED_TONEMAP", fused)
        rr = R.RtRenderer(ctx, scene, opt, (W, H), use_torch=False, tonemap=dict(op=R.TONEMAP_FILMIC, exposure=1.0))
        rr.render()
        first = rr.download("display")
        rr.tonemap.info.exposure = 2.5
        rr.tonemap.info.op = 3
        rr.render()
        second = rr.download("display")
        rr.render(tonemap=False)
        third = rr.download("display")
        rr.render()
        fourth = rr.download("display")
        rr.close()
        shown[fused] = (first, second, third, fourth)
        assert not np.array_equal(first, second)
        assert np.array_equal(third, second), "render(tonemap=False) wrote the display image"
        assert not np.array_equal(fourth, second)
    for k in range(4):
        assert np.array_equal(shown["0"][k].view(np.uint32), shown["1"][k].view(np.uint32)), f"edited tonemap parameters, frame {k}"
    # the direct stage resolves its samples in a kernel of its own: the renderer keeps the tonemap stage, the library says why
    dopt = R.options_for_scene(scene, max_bounces=2, samples_per_pixel=2, samples_per_pass=2)
    monkeypatch.setenv("TRHIP_FUSED_TONEMAP", "1")
    d = R.RtRenderer(ctx, scene, dopt, (W, H), use_torch=False, stage_cls=R.DirectStage)
    assert not d.fused_tonemap
    d.render()
    shown = d.download("display")
    assert np.isfinite(shown).all() and shown[..., :3].max() > 0
    with pytest.raises(RuntimeError, match="direct stage"):
        d.slots[0].pt.set_fused_tonemap(d.slots[0].display, d.tonemap.info)
    d.close()

