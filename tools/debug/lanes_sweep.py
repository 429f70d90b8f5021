"""One frame at a time, a rank's share of the frame by lane count: ms per frame.  usage: lanes_sweep.py [workload]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS
W, H = 1920, 1080
wname = sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"
ctx = R.Context(0)
sc = scenes.WORKLOADS[wname](W, H)
opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
def measure(world, lanes=0, n=96):
    rr = R.RtRenderer(ctx, sc, opt, (W, H), strategy=DISTRIBUTION_SHUFFLED_STRIPS, rank=world - 1, world_size=world, use_torch=False)
    if lanes: rr.slots[0].pt.set_lanes(lanes)
    def frames(n):
        for _ in range(n):
            rr.reset_accumulation(); rr.render_partial(); rr.sync()
    frames(12)
    t0 = time.perf_counter(); frames(n); dt = (time.perf_counter() - t0) / n * 1e3
    rr.close()
    return dt
print(f"# {wname}, TRHIP_FUSED_GRID_FACTOR={os.environ.get('TRHIP_FUSED_GRID_FACTOR', 'default')}: ms per frame, one frame at a time; lanes 0 = the library's choice")
print(f"{'share':>6} " + " ".join(f"{'lanes ' + str(l):>9}" for l in (0, 1, 2, 3, 4)))
for world in (1, 2, 4, 8, 16, 32):
    print(f"{'1/' + str(world):>6} " + " ".join(f"{measure(world, l):>9.3f}" for l in (0, 1, 2, 3, 4)), flush=True)
