"""Host time of the calls of a one-frame-at-a-time frame (python binding): reset_accumulation, render (enqueue), sync (wait)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tauray_amd import renderer as R, scenes
W, H = 1920, 1080
ctx = R.Context(0)
sc = scenes.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"](W, H)
opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
rr = R.RtRenderer(ctx, sc, opt, (W, H), use_torch=False)
for _ in range(10):
    rr.reset_accumulation(); rr.render(); rr.sync()
n = 200
ta = tb = tc = 0.0
t_all = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter(); rr.reset_accumulation(); t1 = time.perf_counter(); rr.render(); t2 = time.perf_counter(); rr.sync(); t3 = time.perf_counter()
    ta += t1 - t0; tb += t2 - t1; tc += t3 - t2
t_all = time.perf_counter() - t_all
print(f"per frame: {t_all / n * 1e3:.4f} ms = reset {ta / n * 1e6:.1f} us + render (enqueue) {tb / n * 1e6:.1f} us + sync (wait) {tc / n * 1e6:.1f} us")
# without the tonemap
for _ in range(n):
    rr.reset_accumulation(); rr.render_partial(); rr.sync()
t0 = time.perf_counter()
for _ in range(n):
    rr.reset_accumulation(); rr.render_partial(); rr.sync()
print(f"path tracing only (render_partial): {(time.perf_counter() - t0) / n * 1e3:.4f} ms")
