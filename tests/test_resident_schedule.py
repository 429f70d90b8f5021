"""The resident schedule (csrc/frame_kernel.h, trhip_pt_set_schedule): a launch whose paths fit on the device at once is rendered by one
kernel that keeps every path in its wave through all bounces.  It has to be the queue schedule's frame bit for bit - in both shading
arithmetics, for the ahead-of-time instances, the general kernels and a program compiled for the option set, with several samples per
pass, accumulated frames, shards and the demodulated targets - because a job may render a full frame one way and its shards the other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from tauray_amd import renderer
    return renderer


@pytest.fixture(scope="module")
def ctx(R):
    return R.Context(0)


def _frames(R, ctx, ss, scene, dist, schedule, ieee, specialize=None, frames=1, targets=("color",), **kw):
    from tauray_amd.distribution import get_distribution_target_size
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, **kw), dist)
    pt.set_schedule(schedule)
    pt.set_shading_arithmetic(ieee)
    if specialize is not None:
        pt.set_specialization(specialize)
    tw, th = get_distribution_target_size(dist)
    bufs = {t: ctx.alloc(tw * th * 16).zero() for t in targets}
    for _ in range(frames):
        pt.run_targets(bufs) if len(targets) > 1 else pt.run(bufs["color"])
    out = {t: b.download((th, tw, 4)) for t, b in bufs.items()}
    c = pt.counters()
    assert c["stack_overflows"] == 0
    pt.close()
    return out, c


def test_resident_schedule_renders_the_frame_of_the_queue_schedule(R, ctx, test_glb_128):
    from tauray_amd import scenes
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE, DISTRIBUTION_SCANLINE, DISTRIBUTION_SHUFFLED_STRIPS
    scene = scenes.test_glb(256, 256)
    ss = R.SceneStage(ctx, scene)
    dup = DistributionParams((256, 256), DISTRIBUTION_DUPLICATE, 0, 1, True)
    cases = [
        ("command-line set", dup, dict(max_bounces=4), {}),
        ("8 bounces, accumulated frames", dup, dict(max_bounces=8), dict(frames=3)),
        ("sobol-owen, compiled program", dup, dict(max_bounces=4, sampler=1), {}),
        ("sobol-z3 + regularisation + clamp, general kernels", dup, dict(max_bounces=5, sampler=3, regularization_gamma=0.2, indirect_clamping=4.0), dict(specialize=False)),
        ("4 spp, 2 per pass, blackman-harris", dup, dict(max_bounces=3, samples_per_pixel=4, samples_per_pass=2, film=2), {}),
        ("scanline shard 3 of 8", DistributionParams((256, 256), DISTRIBUTION_SCANLINE, 3, 8, False), dict(max_bounces=4), {}),
        ("shuffled-strip shard", DistributionParams((256, 256), DISTRIBUTION_SHUFFLED_STRIPS, 1, 3, False), dict(max_bounces=4, russian_roulette_delta=1.5), {}),
        ("hidden lights, white albedo, transparent background", dup, dict(max_bounces=3, hide_lights=1, use_white_albedo_on_first_bounce=1, transparent_background=1), {}),
    ]
    for name, dist, kw, extra in cases:
        for ieee in (True, False):
            q, cq = _frames(R, ctx, ss, scene, dist, 1, ieee, **extra, **kw)
            r, cr = _frames(R, ctx, ss, scene, dist, 2, ieee, **extra, **kw)
            assert np.isfinite(q["color"]).all() and q["color"][..., :3].mean() > 1e-3
            assert np.array_equal(q["color"], r["color"]), f"{name}, ieee={ieee}: {int((q['color'] != r['color']).any(-1).sum())} pixels differ"
            assert cq["closest_rays"] == cr["closest_rays"] and cq["shadow_rays"] == cr["shadow_rays"], (name, ieee)


def test_resident_schedule_at_the_size_of_a_rank_of_eight(R, ctx):
    """BASELINE config 4's scene, the rows one GPU of eight renders (1920 x 135 of 1080): what the schedule is for.  Same bits as the
    queue schedule; the stage picks the resident schedule by itself at this size."""
    from tauray_amd import scenes
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_SCANLINE
    W, H = 1920, 1080
    scene = scenes.sponza_teapots(W, H)
    ss = R.SceneStage(ctx, scene)
    dist = DistributionParams((W, H), DISTRIBUTION_SCANLINE, 7, 8, False)
    q, cq = _frames(R, ctx, ss, scene, dist, 1, False, max_bounces=4)
    r, cr = _frames(R, ctx, ss, scene, dist, 2, False, max_bounces=4)
    a, ca = _frames(R, ctx, ss, scene, dist, 0, False, max_bounces=4)
    assert q["color"].shape == (135, 1920, 4) and q["color"][..., :3].mean() > 1e-3
    assert np.array_equal(q["color"], r["color"]) and np.array_equal(q["color"], a["color"])
    assert cq["closest_rays"] == cr["closest_rays"] == ca["closest_rays"] and cq["shadow_rays"] == cr["shadow_rays"]
