"""BASELINE.json configs 3 and 5 at their real workloads, and the image-L2 numbers north_star asks for.

  * config 3: the Sponza-class scene at its full 251 k triangles, >= 1024 accumulated samples per pixel, 4 bounces - HIP
    against the CPU oracle on the same seeds with the stated bound asserted as written: per-pixel L2 (RMS) of the fp32 radiance
    < 1e-3; plus the full-size property (1920x1080: N accumulated 1-spp frames == one frame of N spp, bit for bit).
  * config 5: the 5x9 light-field grid of `generate_cameras` (src/tauray.cc:680-727: spacing 0.02, recentering distance 5),
    45 x 1080p viewports of the same scene in one launch - layers bit-equal to single-view renders, view-sharded over eight
    ranks == unsharded, and against the oracle at a size it finishes in seconds.
  * L2(HIP, validate_path-tracer.exr): the converged reference image of test.glb against a HIP render of >= 4096 spp through
    the same tonemap - the reference's own acceptance metric (ImageMagick `compare -metric mse`, test/validate_render.py:26-45)
    and its square root.

Numbers are printed (run with -s) and written to gpurun_out/baseline_configs.json so that DESIGN.md can quote them.
TRHIP_TEST_SPP raises the sample counts (default 1024 for the oracle comparison, 4096 where only the GPU works)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden

pytestmark = pytest.mark.gpu

SPP_ORACLE = int(os.environ.get("TRHIP_TEST_SPP", "1024"))
SPP_GPU = max(int(os.environ.get("TRHIP_TEST_SPP", "4096")), 4096)
L2_BOUND = 1e-3      # north_star: "image L2 error < 1e-3", per-pixel L2 on fp32 radiance


def _report(key, values):
    path = os.path.join(ROOT, "gpurun_out", "baseline_configs.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        try:
            data = json.load(open(path))
        except Exception:
            data = {}
    data[key] = values
    json.dump(data, open(path, "w"), indent=1)
    print(f"\n[{key}] " + json.dumps(values))


@pytest.fixture(scope="module")
def R():
    from tauray_amd import renderer
    return renderer


@pytest.fixture(scope="module")
def ctx(R):
    return R.Context(0)


def _dup(size):
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    return DistributionParams(tuple(size), DISTRIBUTION_DUPLICATE, 0, 1, True)


def _rms(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt((d * d).mean()))


def test_config3_sponza_class_accumulated_radiance_l2_vs_oracle(R, ctx, oracle):
    """Config 3's integrand at a size the oracle finishes in about a minute on the box's host cores: the full 251 k-triangle
    scene, SPP_ORACLE samples per pixel accumulated by the running mean of gbuffer.glsl:18-28, 4 bounces, same seeds."""
    from tauray_amd import scenes
    W, H, N = 320, 180, SPP_ORACLE
    scene = scenes.sponza_class(width=W, height=H)
    assert scene.triangle_count > 250_000
    ss = R.SceneStage(ctx, scene)
    kw = dict(max_bounces=4, samples_per_pixel=N, samples_per_pass=1)
    ref = oracle.OracleScene(scene).render_pt(oracle.options_for_scene(scene, **kw), W, H)[0]
    # Both arithmetic modes of the shading kernel (include/trhip.h, trhip_pt_set_shading_arithmetic): IEEE fp32 like the oracle, and
    # the default - what Vulkan asks of the reference's GLSL (csrc/shade_fast.hip).  The BASELINE bound holds for either.
    for mode, ieee in (("ieee", True), ("default", False)):
        pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, **kw), _dup((W, H)))
        pt.set_shading_arithmetic(ieee)
        color = ctx.alloc(W * H * 16).zero()
        pt.run(color)
        img = color.download((H, W, 4))
        assert pt.counters()["stack_overflows"] == 0
        pt.close()
        # the integrand has a NaN sample about once in 8e7 and a running mean keeps it: few such pixels, and
        # at IEEE fp32 the same ones in both images (inf / inf needs the same overflow); the L2 figures are over the others
        nan_ref, nan_img = np.isnan(ref[..., :3]).any(-1), np.isnan(img[..., :3]).any(-1)
        assert nan_ref.sum() <= 8 and nan_img.sum() <= 8 and not np.isinf(img).any() and not np.isinf(ref).any()
        if ieee:
            assert np.array_equal(nan_img, nan_ref)
        ok = ~(nan_ref | nan_img)
        rgb, rrgb = img[..., :3][ok], ref[..., :3][ok]
        rms = _rms(rgb, rrgb)
        per_pixel = np.sqrt(((rgb.astype(np.float64) - rrgb) ** 2).sum(-1))        # L2 norm of the pixel's rgb difference
        _report("config3_hip_vs_oracle" + ("" if ieee else "_default_shading_arithmetic") + ("" if N == 1024 else f"_{N}spp"), {
            "scene": "sponza_class", "triangles": int(scene.triangle_count), "size": [W, H], "spp": N, "bounces": 4, "shading_arithmetic": mode,
            "nan_pixels": [int(nan_img.sum()), int(nan_ref.sum())],
            "rms_radiance": rms, "mse_radiance": rms * rms, "max_pixel_l2": float(per_pixel.max()), "p999_pixel_l2": float(np.quantile(per_pixel, 0.999)),
            "bit_equal_pixels": float((rgb == rrgb).all(-1).mean()), "mean_radiance": float(rrgb.mean()), "max_radiance": float(rrgb.max()),
            "mean_rel_err": abs(float(rgb.mean()) - float(rrgb.mean())) / float(rrgb.mean())})
        assert rms < L2_BOUND, f"{mode}: RMS radiance error {rms:.3e} vs the oracle"
        assert float(np.quantile(per_pixel, 0.999)) < 10 * L2_BOUND
        assert abs(float(rgb.mean()) - float(rrgb.mean())) / float(rrgb.mean()) < (1e-4 if ieee else 1e-3)
        assert np.array_equal(img[..., 3], ref[..., 3])


def test_config3_full_size_accumulation_property(R, ctx):
    """1920x1080, 4 bounces, SPP_GPU samples: accumulating 1-spp frames (the sample counter advances frame by frame,
    src/rt_stage.cc:81) is bit for bit the same image as one frame of SPP_GPU passes (control.previous_samples advances,
    src/path_tracer_stage.cc:118-147) - the two ways the reference reaches 4096 spp; finite, alpha 1, and converged enough
    that two disjoint halves of the samples agree."""
    from tauray_amd import scenes
    W, H, N = 1920, 1080, SPP_GPU
    scene = scenes.sponza_class(width=W, height=H)
    ss = R.SceneStage(ctx, scene)
    acc = ctx.alloc(W * H * 16).zero()
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4), _dup((W, H)))
    import time
    half = None
    ctx.sync()
    t0 = time.perf_counter()
    for f in range(N):
        pt.run(acc)
        if f == N // 2 - 1:
            half = acc.download((H, W, 4))
    frames = acc.download((H, W, 4))
    seconds_frames = time.perf_counter() - t0
    assert pt.counters()["stack_overflows"] == 0
    pt.close()
    one = ctx.alloc(W * H * 16).zero()
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4, samples_per_pixel=N, samples_per_pass=1), _dup((W, H)))
    ctx.sync()
    t0 = time.perf_counter()
    pt.run(one)              # BASELINE config 3 as the reference renders it: one frame, N one-sample passes (sample lanes)
    ctx.sync()
    seconds_passes = time.perf_counter() - t0
    rays = pt.counters()
    rays = rays["closest_rays"] + rays["shadow_rays"]
    passes = one.download((H, W, 4))
    pt.close()
    # The integrand itself produces a NaN sample about once in 40 frames of this scene - in the oracle at the same pixel of the
    # same frame (tools/find_nonfinite.py) - and the running mean of gbuffer.glsl:18-28 keeps it: those
    # pixels must agree between the two schedules like all others, and there must be few of them.
    nan_px = np.isnan(frames[..., :3]).any(-1)
    ok = ~nan_px
    assert nan_px.mean() < 2e-4, f"{int(nan_px.sum())} NaN pixels"
    assert not np.isinf(frames).any() and (frames[..., 3] == 1).all() and frames[..., :3][ok].min() >= 0
    assert np.array_equal(frames, passes, equal_nan=True), f"{int((frames != passes).any(-1).sum())} pixels differ between {N} frames and {N} passes"
    # second half of the samples = 2 * mean(all) - mean(first half): Monte-Carlo noise of an N/2-sample image, for scale
    second = 2.0 * frames[..., :3][ok].astype(np.float64) - half[..., :3][ok]
    noise = _rms(half[..., :3][ok], second)
    _report("config3_full_size", {"scene": "sponza_class", "size": [W, H], "spp": N, "frames_equal_passes": True, "nan_pixels": int(nan_px.sum()),
                                  "rms_between_sample_halves": noise, "mean_radiance": float(frames[..., :3][ok].mean()),
                                  "seconds_one_frame_of_passes": seconds_passes, "ms_per_sample": seconds_passes / N * 1e3,
                                  "mray_per_s": rays / seconds_passes / 1e6, "seconds_accumulated_one_sample_frames": seconds_frames})


# What the primary ray hits in test/test.glb (instance ids of the feature renderer) and how the HIP image may differ from the golden
# there: (largest relative offset of the region's mean, largest RMS of its 16x16 block means), each more than twice what
# tools/golden_residual.py measures at 16 384 spp (profiles/r3/golden_residual.json).  The per-pixel difference is the golden's own
# Monte-Carlo noise (0.031 RMS; this render's is 0.010); block means average the noise of both images out, and what is left is
# systematic - and has names:
#   * torus: the checkout adds the emission of a directly visible emitter twice (shader/path_tracer.glsl:421-435 +
#     path_tracer.rgen:112); the golden shows it once: +13 %.
#   * teapot (metal, albedo (0.8, 0.077, 0)): the checkout's material model weighs the metallic lobe with the albedo alone
#     (modulate_bsdf, shader/material.glsl:52-55; ggx_brdf_inner adds `geometry * distribution * cos_l * metallic` without a Fresnel
#     term, shader/ggx.glsl:145-146), so a metal with no blue in its albedo reflects no blue at any angle: the blue channel of every
#     teapot pixel is exactly 0 here and in the oracle.  The golden's teapot has white highlights (blue up to 0.38, above 0.01 in a
#     tenth of its pixels): it was rendered by an earlier material model whose metals go white at grazing angles.  -6.5 %.
#   * Suzanne (glass): the same earlier model on the transmission lobe (albedo * transmission today): -4.8 %.
#   * room faces: +0.2 ... +2.3 %, largest on the darkest wall (mean 0.07), whose light is all indirect off the objects above.
#   * plane (alpha-blended): -0.2 %.
# Round 5: every region's mean is held to a WINDOW around its measured, explained residual (16 384 spp, seeds fixed: the residual is a number,
# not a random variable), +- the larger of 0.4 % and three tenths of the residual - a bound of "less than two or three times the residual"
# let a 5 % energy error in the metallic or the transmission lobe pass (teapot -6.5 % had a bound of 14 %); the windows do not
# (teapot: -8.4 ... -4.5 %).  Block means: 1.5 x the measured figure.
#        region: (name, measured relative offset of the mean, its window half-width, bound on the 16 x 16 block RMS)
GOLDEN_REGIONS = {0: ("room face 0", 0.0113, 0.004, 0.0122), 1: ("room face 1", 0.0018, 0.004, 0.0059), 2: ("room face 2", 0.0229, 0.0069, 0.0047),
                  3: ("room face 3", 0.0022, 0.004, 0.0152), 4: ("teapot", -0.0648, 0.0194, 0.0353), 5: ("suzanne", -0.0479, 0.0144, 0.044),
                  6: ("torus", 0.1345, 0.04, 0.0), 7: ("plane", -0.0020, 0.004, 0.0092)}


def test_mse_and_region_means_against_the_reference_golden_image(R, ctx):
    """validate_path-tracer.exr (test/references, 512x512 half, filmic + gamma 2.2) is the one image of the Vulkan path tracer
    that exists here.  HIP render with the CLI defaults it was made with (8 bounces, uniform-random sampler, point film) at
    SPP_GPU * 4 samples per pixel, same tonemap, compared
      * as the reference's own test does: the MEAN SQUARED error over the image (ImageMagick `compare -metric mse`, test/validate_render.py:
        26-45) - this is the figure held to 1e-3 below, and it is an MSE: its square root, the RMS (per-pixel L2), is 0.031 and is the
        golden image's own one-sample-per-pixel-scale noise (this render's is 0.010), not an error of 3 %;
      * region by region (GOLDEN_REGIONS above): the mean of every region inside a window around its explained residual, block means
        bounded - the part of the comparison that says something about the integrator.
    "Image L2 < 1e-3 against the Vulkan reference" of north_star cannot be shown against this golden, which predates the checkout's material
    model, nor against a live Vulkan render (no ICD here): it is shown against the oracle (config 3: RMS 1.3e-5)."""
    from tauray_amd.gltf import load_glb
    W = H = 512
    N = SPP_GPU * 4
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    ss = R.SceneStage(ctx, scene)
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, samples_per_pixel=N, samples_per_pass=1), _dup((W, H)))
    color, disp = ctx.alloc(W * H * 16).zero(), ctx.alloc(W * H * 16)
    pt.run(color)
    R.TonemapStage(ctx).run(color, disp, W, H)
    ours = disp.download((H, W, 4))[..., :3]
    pt.close()
    fs = R.FeatureStage(ctx, ss, 9, _dup((W, H)))
    color.zero()
    fs.run(color)
    ids = color.download((H, W, 4))[..., 0]
    ids = np.where(np.isnan(ids), -1, ids).astype(np.int32)
    gold = load_golden("path-tracer")
    assert np.isfinite(ours).all()
    torus = ids == 6
    assert torus.sum() > 4000 and np.abs(gold[torus][:, 0] - 0.8413).mean() < 0.02      # the visible emitter: where the id image says
    keep = ~torus
    d2 = ((ours.astype(np.float64) - gold) ** 2)
    mse_all, mse_keep = float(d2.mean()), float(d2[keep].mean())
    b = 16

    def block_rms(mask):      # RMS over the 16x16 blocks (with at least 64 pixels of the region) of the difference of the block means
        bo = np.where(mask[..., None], ours, 0).reshape(H // b, b, W // b, b, 3).sum((1, 3))
        bg = np.where(mask[..., None], gold, 0).reshape(H // b, b, W // b, b, 3).sum((1, 3))
        n = mask.reshape(H // b, b, W // b, b).sum((1, 3))
        valid = n >= 64
        return float(np.sqrt((((bo - bg) / np.maximum(n, 1)[..., None])[valid] ** 2).mean()))

    regions = {}
    for k, (name, residual, half_width, max_block) in GOLDEN_REGIONS.items():
        m = ids == k
        assert m.sum() > 4000, name
        off = float((ours[m].mean() - gold[m].mean()) / gold[m].mean())
        regions[name] = {"pixels": int(m.sum()), "rel_mean_offset": off, "block16_rms": block_rms(m), "window": [residual - half_width, residual + half_width]}
        assert abs(off - residual) < half_width, f"{name}: mean differs by {off:+.2%}, outside {residual - half_width:+.2%} ... {residual + half_width:+.2%}"
        if max_block:
            assert regions[name]["block16_rms"] < max_block, f"{name}: block RMS {regions[name]['block16_rms']:.4f} (bound {max_block})"
    # the two differences with a known cause have the sign and size that cause gives them
    assert 0.08 < regions["torus"]["rel_mean_offset"] < 0.20
    teapot = ids == 4
    assert float(ours[teapot][:, 2].max()) == 0.0 and float((gold[teapot][:, 2] > 0.01).mean()) > 0.05 and regions["teapot"]["rel_mean_offset"] < -0.03
    _report("hip_vs_reference_golden", {
        "image": "validate_path-tracer.exr", "spp": N, "mse_all_pixels": mse_all, "rms_all_pixels": mse_all ** 0.5,
        "mse_without_visible_emitter": mse_keep, "rms_without_visible_emitter": mse_keep ** 0.5, "block16_rms": block_rms(keep),
        "emitter_pixels": int(torus.sum()), "mean_rel_err": abs(float(ours[keep].mean()) - float(gold[keep].mean())) / float(gold[keep].mean()),
        "regions": regions})
    assert mse_keep < L2_BOUND, f"MSE {mse_keep:.3e} against the reference image"      # a mean SQUARED error (the reference's metric); 9.4e-4 of it is the golden's own noise
    assert block_rms(keep) < 0.02
    assert mse_all < 0.15       # the reference's own tolerance: 10000 on ImageMagick's Q16 scale (test/CMakeLists.txt)


def test_config4_full_size_frame_vs_oracle(R, ctx, oracle):
    """BASELINE config 4's frame at the size bench.py times it: sponza_teapots (1 M triangles), 1920 x 1080, 1 spp, 4 bounces - the oracle
    renders it in a few seconds on the box's host cores.  Two frame indices (two sets of random streams), both shading arithmetics: IEEE fp32
    against the oracle pixel by pixel (it follows the oracle expression by expression: bit-equal in all but a handful of pixels), the default
    arithmetic within the per-pixel tolerance.  tests/test_gpu_parity.py compares the same scene at 480 x 272; this is the frame of the headline."""
    from tauray_amd import scenes
    W, H = 1920, 1080
    scene = scenes.sponza_teapots(width=W, height=H)
    assert scene.triangle_count > 900_000
    ss = R.SceneStage(ctx, scene)
    osc = oracle.OracleScene(scene)
    kw = dict(max_bounces=4)
    out = {}
    for frame in (0, 3):
        ref = osc.render_pt(oracle.options_for_scene(scene, **kw), W, H, frame_counter=frame)[0]
        for mode, ieee in (("ieee", True), ("default", False)):
            pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, **kw), _dup((W, H)))
            pt.set_shading_arithmetic(ieee)
            pt.set_frame_counter(frame)
            color = ctx.alloc(W * H * 16).zero()
            pt.run(color)
            img = color.download((H, W, 4))
            assert pt.counters()["stack_overflows"] == 0
            pt.close()
            assert np.isfinite(img).all()
            a, b = img[..., :3].astype(np.float64), ref[..., :3].astype(np.float64)
            rel = np.abs(a - b) / (np.abs(b) + 1e-2)
            outside = float((rel.max(-1) > 1e-2).mean())
            bit_equal = float((img[..., :3] == ref[..., :3]).all(-1).mean())
            mean_rel = abs(a.mean() - b.mean()) / b.mean()
            out[f"frame {frame} {mode}"] = {"pixels_outside_1e-2": outside, "bit_equal_pixels": bit_equal, "mean_rel_err": float(mean_rel), "rms": _rms(a, b)}
            assert outside < (2e-4 if ieee else 5e-3), f"frame {frame}, {mode}: {outside:.4%} of the pixels outside 1 %"
            assert mean_rel < (1e-4 if ieee else 2e-3), f"frame {frame}, {mode}: image mean off by {mean_rel:.3e}"
            if ieee:
                assert bit_equal > 0.5, f"frame {frame}: only {bit_equal:.4%} of the pixels bit-equal to the oracle at IEEE fp32"      # 74 % measured: the rest differ in the last bits (libm against the device's sin / cos / pow)
            assert np.array_equal(img[..., 3], ref[..., 3])
    _report("config4_full_size_hip_vs_oracle", {"scene": "sponza_teapots", "triangles": int(scene.triangle_count), "size": [W, H], "bounces": 4, "spp": 1, **out})


def test_config5_light_field_grid_full_size(R, ctx):
    """Config 5: 45 viewports (5 rows x 9 columns, spacing 0.02, recentering distance 5) of the Sponza-class scene at
    1920x1080, 1 spp, 4 bounces, as ONE launch with gl_LaunchIDEXT.z = 45 (src/rt_camera_stage.cc:162-166).  Layers of the
    batch are bit-equal to single-view renders of the same global viewport, and the eight view shards of an 8-GPU job
    (viewport v on rank v mod 8) hold exactly the layers of the unsharded frame."""
    from tauray_amd import scenes
    from tauray_amd.scene import generate_camera_grid
    W, H, V = 1920, 1080, 45
    scene = scenes.sponza_class(width=W, height=H)
    scene.cameras = generate_camera_grid(scene.cameras[0], 9, 5, 0.02, 0.02, 5.0)
    assert len(scene.cameras) == V
    opt = R.options_for_scene(scene, max_bounces=4)
    full = R.RtRenderer(ctx, scene, opt, (W, H), viewports=V, use_torch=False)
    full.render(tonemap=False)
    batch = full.download("color")
    counters = full.counters()
    assert counters["stack_overflows"] == 0
    ss = full.scene_update
    assert batch.shape == (V, H, W, 4) and np.isfinite(batch).all() and (batch[..., 3] == 1).all()
    assert not np.array_equal(batch[0], batch[44]) and not np.array_equal(batch[21], batch[22])
    # single-view renders of global viewports 0, 4, 22, 40, 44 (corners and centre of the grid)
    for v in (0, 4, 22, 40, 44):
        pt = R.PathTracerStage(ctx, ss, opt, _dup((W, H)))
        pt.set_shard(viewport_base=v, viewport_stride=1)
        buf = ctx.alloc(W * H * 16).zero()
        pt.run(buf, 1)
        one = buf.download((H, W, 4))
        pt.close()
        assert np.array_equal(one, batch[v]), f"viewport {v}: {int((one != batch[v]).any(-1).sum())} pixels differ from the batched layer"
    # eight view shards on fake devices (one Context per rank, as the C++ host's --fake-devices): 6,6,6,6,6,5,5,5 views
    sizes = []
    for rank in range(8):
        c = R.Context(0)
        rr = R.RtRenderer(c, scene, opt, (W, H), rank=rank, world_size=8, viewports=V, use_torch=False, shard="views")
        rr.render(tonemap=False)
        mine = rr.download("color")
        sizes.append(mine.shape[0])
        assert np.array_equal(mine, batch[rank::8]), f"rank {rank}: its views differ from the unsharded layers"
        rr.close()
        c.close()
    assert sizes == [6, 6, 6, 6, 6, 5, 5, 5]
    _report("config5_full_size", {"scene": "sponza_class", "views": V, "size": [W, H], "rays": counters["closest_rays"] + counters["shadow_rays"],
                                  "single_views_checked": [0, 4, 22, 40, 44], "view_shards": sizes})
    full.close()


def test_config5_light_field_grid_vs_oracle(R, ctx, oracle):
    """The same 45-camera grid at 96x54 against the oracle (every layer), radiance within the fp32 tolerance of the parity
    tests and primary hit distances bit-equal for all 45 cameras."""
    from tauray_amd import scenes
    from tauray_amd.scene import generate_camera_grid
    from tauray_amd.distribution import get_distribution_target_size
    W, H, V = 96, 54, 45
    scene = scenes.sponza_class(width=W, height=H)
    scene.cameras = generate_camera_grid(scene.cameras[0], 9, 5, 0.02, 0.02, 5.0)
    ss = R.SceneStage(ctx, scene)
    osc = oracle.OracleScene(scene)
    kw = dict(max_bounces=4)
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, **kw), _dup((W, H)))
    color = ctx.alloc(V * W * H * 16).zero()
    pt.run(color, V)
    img = color.download((V, H, W, 4))
    pt.close()
    ref = osc.render_pt(oracle.options_for_scene(scene, **kw), W, H, viewports=V)
    rel = np.abs(img[..., :3] - ref[..., :3]) / (np.abs(ref[..., :3]) + 1e-2)
    bad = float((rel.max(-1) > 1e-2).mean())
    assert bad <= 2e-3, f"{bad:.4%} pixels differ"
    assert abs(float(img[..., :3].mean()) - float(ref[..., :3].mean())) / float(ref[..., :3].mean()) < 2e-3
    for v in (0, 8, 22, 36, 44):
        fs = R.FeatureStage(ctx, ss, 5, _dup((W, H)))
        buf = ctx.alloc(W * H * 16).zero()
        fs.run(buf, viewport=v)
        g, r = buf.download((H, W, 4)), osc.render_feature(5, W, H, viewport=v)
        assert not (~((g == r) | (np.isnan(g) & np.isnan(r)))).any(), f"hit distances of camera {v}"
    _report("config5_vs_oracle", {"views": V, "size": [W, H], "pixels_outside_1e-2": bad, "bit_equal_pixels": float((img[..., :3] == ref[..., :3]).all(-1).mean())})
