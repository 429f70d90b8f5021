"""The numerator of the benchmark's metric (BASELINE.json: Mray/s = closest-hit + shadow rays actually traced per second,
SURVEY.md section 8(d)) is a device counter: these tests pin it to the CPU oracle's own count of the same frame.

Paths are the same paths in both implementations up to a handful of discrete decisions that libm ulps flip (DESIGN.md section 3:
`sinf / cosf / powf` differ between glibc and ocml), so primary rays agree exactly and the totals to 1e-4 relative."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _dup(size):
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    return DistributionParams(tuple(size), DISTRIBUTION_DUPLICATE, 0, 1, True)


def _hip_counters(R, ctx, ss, scene, size, frames, ieee=False, **kw):
    W, H = size
    opt = R.options_for_scene(scene, **kw)
    out = {}
    for counting in (False, True):        # the production kernels count rays; the counting instances also count the work
        pt = R.PathTracerStage(ctx, ss, opt, _dup(size))
        pt.set_shading_arithmetic(ieee)
        pt.set_profiling(counting, False)
        buf = ctx.alloc(W * H * 16).zero()
        for _ in range(frames):
            pt.reset_accumulated_samples()
            pt.run(buf)
        out[counting] = pt.counters()
        pt.close()
    assert out[False]["stack_overflows"] == 0
    for k in ("closest_rays", "shadow_rays"):
        assert out[False][k] == out[True][k], k
    return out[True]


def _oracle_counters(oracle, osc, scene, size, frames, **kw):
    W, H = size
    oopt = oracle.options_for_scene(scene, **kw)
    osc.reset_counters()
    for f in range(frames):
        osc.render_pt(oopt, W, H, frame_counter=f)
    return osc.counters()


def _check(hip, ora, pixels, frames, what, tol=1e-4):
    # the first closest-hit ray of every pixel is traced unconditionally by both
    assert hip["closest_rays"] >= pixels * frames and ora["closest_rays"] >= pixels * frames
    for k in ("closest_rays", "shadow_rays", "surface_hits"):
        assert ora[k] > 0, (what, k)
        rel = abs(hip[k] - ora[k]) / ora[k]
        assert rel <= tol, f"{what}: {k} {hip[k]} (HIP) vs {ora[k]} (oracle): {rel:.2e} relative"


@pytest.mark.gpu
def test_ray_counters_match_the_oracle_on_test_glb(oracle):
    from tauray_amd import renderer as R
    from tauray_amd.gltf import load_glb
    W = H = 128
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, scene)
    osc = oracle.OracleScene(scene)
    # one bounce: every ray of the frame is a primary ray or its shadow ray - the primary count is exact
    hip1 = _hip_counters(R, ctx, ss, scene, (W, H), 1, max_bounces=1)
    ora1 = _oracle_counters(oracle, osc, scene, (W, H), 1, max_bounces=1)
    assert hip1["closest_rays"] == ora1["closest_rays"] == W * H
    ora = _oracle_counters(oracle, osc, scene, (W, H), 4, max_bounces=4)
    # shading at IEEE fp32 like the oracle: 1e-4; with the default shading arithmetic (csrc/shade_fast.hip) a few more paths flip
    # a discrete decision (a shadow ray cast or not, a path continued or not): 5e-4
    _check(_hip_counters(R, ctx, ss, scene, (W, H), 4, ieee=True, max_bounces=4), ora, W * H, 4, "test.glb 128x128, 4 bounces, 4 frames, IEEE shading")
    _check(_hip_counters(R, ctx, ss, scene, (W, H), 4, max_bounces=4), ora, W * H, 4, "test.glb 128x128, 4 bounces, 4 frames", tol=5e-4)


@pytest.mark.gpu
def test_ray_counters_match_the_oracle_on_the_bench_scene(oracle):
    """BASELINE config 4's scene (sponza_teapots, 1 M triangles) as bench.py renders it, at 160 x 90."""
    from tauray_amd import renderer as R
    from tauray_amd import scenes
    W, H = 160, 90
    scene = scenes.sponza_teapots(width=W, height=H)
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, scene)
    osc = oracle.OracleScene(scene)
    hip1 = _hip_counters(R, ctx, ss, scene, (W, H), 1, max_bounces=1)
    ora1 = _oracle_counters(oracle, osc, scene, (W, H), 1, max_bounces=1)
    assert hip1["closest_rays"] == ora1["closest_rays"] == W * H
    frames = 8
    ora = _oracle_counters(oracle, osc, scene, (W, H), frames, max_bounces=4)
    _check(_hip_counters(R, ctx, ss, scene, (W, H), frames, ieee=True, max_bounces=4), ora, W * H, frames, "sponza_teapots 160x90, 4 bounces, 8 frames, IEEE shading")
    hip = _hip_counters(R, ctx, ss, scene, (W, H), frames, max_bounces=4)
    _check(hip, ora, W * H, frames, "sponza_teapots 160x90, 4 bounces, 8 frames", tol=5e-4)
    # and the metric's numerator per frame at this size is what the bench divides by the frame time
    assert abs((hip["closest_rays"] + hip["shadow_rays"]) - (ora["closest_rays"] + ora["shadow_rays"])) <= 5e-4 * (ora["closest_rays"] + ora["shadow_rays"])
