"""What one rank of an N-GPU pixel-sharded job renders per frame - the path tracing of its share alone, nothing exchanged - under the three
definitions bench.py reports: one frame at a time (`value`), two frames in flight with one frame per launch (the reference's own
MAX_FRAMES_IN_FLIGHT = 2, src/context.hh:26: `value_two_in_flight`) and four frame slots of two frames per launch (`value_pipelined`).
The ratio of the whole frame's time to a share's time is the scaling a job of N such ranks can reach before its transport costs anything:
the `scaling_expected_vs_one_gpu` of the bench line.
usage: python tools/shard_share_probe.py [workload] [strategy: 1 scanline | 2 shuffled strips] > profiles/r5/shard_share_probe_<workload>.txt"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS
W, H = 1920, 1080
wname = sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"
strategy = int(sys.argv[2]) if len(sys.argv) > 2 else DISTRIBUTION_SHUFFLED_STRIPS
ctx = R.Context(0)
sc = scenes.WORKLOADS[wname](W, H)
opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
MODES = (("one frame at a time", 1, 1, True), ("two in flight, one frame per launch", 2, 1, False), ("four slots of two frames", 4, 2, False))
table = {}
for world in (1, 2, 4, 8):
    for name, F, B, sync_each in MODES:
        rr = R.RtRenderer(ctx, sc, opt, (W, H), strategy=strategy, rank=world - 1, world_size=world, use_torch=False, frames_in_flight=F, frames_per_launch=B)

        def frames(n):
            for _ in range((n + B - 1) // B):
                rr.reset_accumulation()
                rr.render_partial()
                if sync_each:
                    rr.sync()
            rr.sync()
        frames(12)
        n = 96
        t0 = time.perf_counter(); frames(n); dt = (time.perf_counter() - t0) / n * 1e3
        table[(world, name)] = dt
        rr.close()
print(f"# {wname} {W}x{H}, 4 bounces, 1 spp; the last rank's share of a job of N ranks ({'shuffled strips, equal shares' if strategy == DISTRIBUTION_SHUFFLED_STRIPS else 'scanlines'}); ms per frame")
print(f"{'share':>8} " + " ".join(f"{name:>38}" for name, *_ in MODES))
for world in (1, 2, 4, 8):
    print(f"{'1/' + str(world):>8} " + " ".join(f"{table[(world, name)]:>30.3f} (x{table[(1, name)] / table[(world, name)]:>4.2f})" for name, *_ in MODES))
print("# (xN.NN): the whole frame's time over the share's = the scaling N ranks can reach under that definition before the transport")
