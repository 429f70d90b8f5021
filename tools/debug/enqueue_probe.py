"""Host time to enqueue one frame (trhip_pt_render returns before the GPU finishes) vs GPU frame time."""
import sys, time, os
sys.path.insert(0, '.')
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
ctx = R.Context(0)
for (W, H) in ((1920, 1080), (1920, 136)):
    sc = scenes.WORKLOADS["test_glb"](W, H)
    ss = R.SceneStage(ctx, sc)
    opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
    pt = R.PathTracerStage(ctx, ss, opt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    color = ctx.alloc(W * H * 16).zero()
    for _ in range(5): pt.run(color)
    ctx.sync()
    enq = []
    t0 = time.perf_counter()
    for _ in range(50):
        a = time.perf_counter(); pt.reset_accumulated_samples(); pt.run(color); enq.append(time.perf_counter() - a)
    ctx.sync()
    tot = time.perf_counter() - t0
    enq.sort()
    print(f"{W}x{H} lanes={os.environ.get('TRHIP_LANES','4')}: enqueue p50 {enq[25]*1e3:.3f} ms, frame {tot/50*1e3:.3f} ms")
    pt.close()
