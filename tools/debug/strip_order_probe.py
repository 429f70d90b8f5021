import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS
W, H = 1920, 1080
ctx = R.Context(0)
sc = scenes.WORKLOADS["sponza_teapots"](W, H)
opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
def measure(world, F=1, B=1, n=96, sync_each=True, lanes=0):
    rr = R.RtRenderer(ctx, sc, opt, (W, H), strategy=DISTRIBUTION_SHUFFLED_STRIPS, rank=world - 1, world_size=world, use_torch=False, frames_in_flight=F, frames_per_launch=B)
    if lanes: rr.slots[0].pt.set_lanes(lanes)
    def frames(n):
        for _ in range((n + B - 1) // B):
            rr.reset_accumulation(); rr.render_partial()
            if sync_each: rr.sync()
        rr.sync()
    frames(12)
    t0 = time.perf_counter(); frames(n); dt = (time.perf_counter() - t0) / n * 1e3
    rr.close()
    return dt
def row(tag):
    print(tag, " ".join(f"lanes {l}: {measure(8, lanes=l):.3f}" for l in (0, 1, 2, 4)), flush=True)
row("1/8 first           ")
print("1/1 sync", measure(1))
row("1/8 after 1/1 sync   ")
print("1/1 two in flight", measure(1, 2, 1, sync_each=False))
row("1/8 after 2 in flight")
print("1/1 four slots", measure(1, 4, 2, sync_each=False))
row("1/8 after four slots ")
print("after four slots: 1/1 sync", measure(1), " 1/2", measure(2), " 1/4", measure(4), " 1/1 two in flight", measure(1, 2, 1, sync_each=False), " four slots", measure(1, 4, 2, sync_each=False), flush=True)
row("1/8 again            ")
