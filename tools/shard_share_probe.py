"""What one rank of an N-GPU scanline-sharded job renders per frame: path tracing of rows y = r (mod N) only, with F frames
in flight.  usage: python tools/shard_share_probe.py [workload]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DISTRIBUTION_SCANLINE
W, H = 1920, 1080
wname = sys.argv[1] if len(sys.argv) > 1 else "test_glb"
strategy = int(sys.argv[2]) if len(sys.argv) > 2 else DISTRIBUTION_SCANLINE   # 1 scanline, 2 shuffled strips
ctx = R.Context(0)
sc = scenes.WORKLOADS[wname](W, H)
opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
for world in (1, 2, 4, 8):
    for F in (1, 4, 6, 8):
        rr = R.RtRenderer(ctx, sc, opt, (W, H), strategy=strategy, rank=world - 1, world_size=world, use_torch=False, frames_in_flight=F)
        def frames(n):
            for _ in range(n):
                rr.render_partial()
            rr.sync()
        frames(6)
        t0 = time.perf_counter(); frames(60); dt = (time.perf_counter() - t0) / 60 * 1e3
        print(f"{wname} 1/{world} of the rows, {F} frame(s) in flight: {dt:.3f} ms/frame  (x{world} = {dt * world:.2f})")
        rr.close()
