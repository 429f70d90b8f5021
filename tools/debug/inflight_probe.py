"""Throughput with one vs two frames in flight (two path_tracer_stage objects on two streams)."""
import sys, time
sys.path.insert(0, '.')
import torch
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
ctx = R.Context(0)
import os
W, H = 1920, int(os.environ.get('PROBE_H', '1080'))
for wname in ("test_glb", "sponza_teapots"):
    sc = scenes.WORKLOADS[wname](W, H)
    ss = R.SceneStage(ctx, sc)
    opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
    d = DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True)
    pts = [R.PathTracerStage(ctx, ss, opt, d) for _ in range(3)]
    cols = [ctx.alloc(W * H * 16).zero() for _ in range(3)]
    streams = [torch.cuda.Stream(device=0) for _ in range(3)]
    raw = [s.cuda_stream for s in streams]
    def run(n_streams, frames):
        for f in range(frames):
            k = f % n_streams
            pts[k].reset_accumulated_samples()
            pts[k].run(cols[k], stream=raw[k])
        torch.cuda.synchronize()
    for n_streams in (1, 2, 3):
        run(n_streams, 6)
        t0 = time.perf_counter(); run(n_streams, 40); dt = time.perf_counter() - t0
        print(f"{wname:15s} frames in flight {n_streams}: {dt / 40 * 1e3:.3f} ms/frame")
    for p in pts: p.close()
