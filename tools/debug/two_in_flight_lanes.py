"""Two frames in flight (the reference's MAX_FRAMES_IN_FLIGHT), one frame per launch: lanes per slot.  ms per frame."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tauray_amd import renderer as R, scenes
W, H = 1920, 1080
ctx = R.Context(0)
for wname in sys.argv[1:] or ["sponza_teapots", "sponza_class", "test_glb"]:
    sc = scenes.WORKLOADS[wname](W, H)
    opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
    def measure(F, lanes, n=96):
        rr = R.RtRenderer(ctx, sc, opt, (W, H), use_torch=False, frames_in_flight=F)
        for s in rr.slots:
            s.pt.set_lanes(lanes)
        def frames(n):
            for _ in range(n):
                rr.reset_accumulation(); rr.render_partial()
            rr.sync()
        frames(12)
        t0 = time.perf_counter(); frames(n); dt = (time.perf_counter() - t0) / n * 1e3
        rr.close()
        return dt
    for F in (2, 3, 4):
        print(wname, f"{F} in flight:", " ".join(f"lanes {l}: {measure(F, l):.3f}" for l in (1, 2, 4)), flush=True)
