"""Does an idle extra stream slow a four-lane frame down?  (tools/debug/strip_order_probe.py found: yes once a fifth stream exists.)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS
W, H = 1920, 1080
ctx = R.Context(0)
sc = scenes.WORKLOADS["sponza_teapots"](W, H)
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
def row(tag):
    print(tag, " ".join(f"1/{w} lanes {l}: {measure(w, l):.3f}" for w, l in ((1, 4), (1, 3), (1, 2), (8, 4), (8, 3), (8, 2))), flush=True)
row("clean          ")
buf = ctx.alloc(1 << 20)
extra = []
for k in range(1, 7):
    s = ctx.create_stream()       # the pool is empty: a new stream
    import ctypes as C
    from tauray_amd import _lib
    _lib.lib().trhip_memset(ctx.h, C.c_void_p(buf.ptr), 0, 1 << 20, C.c_void_p(s))
    ctx.sync(s)
    extra.append(s)
    row(f"{k} extra streams")
