"""Where the clocks of a shade wave go, segment by segment (csrc/shade_timeline.h).
usage (on a GPU box):  TRHIP_LIB=tauray_amd/libtrhip_shadetl.so python tools/shade_timeline.py [workload] [world] [frames]
The library is the variant built by
  make -C tauray_amd/csrc variant NAME=shadetl EXTRA=-DTR_SHADE_TIMELINE=1 FASTEXTRA=-DTR_SHADE_TIMELINE=1      (=2: stamps without waits)
world = 1: the whole 1920x1080 frame; world = 8: the last rank's share of a job of 8 ranks (shuffled strips), one frame at a time -
the case in which a shade launch lasts as long as one wave's dependent chain."""
import ctypes as C
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tauray_amd import renderer as R, scenes, _lib   # noqa: E402
from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
W, H = 1920, 1080
lib = _lib.lib()
if not hasattr(lib, "trhip_debug_shade_timeline"):
    sys.exit("this library has no shade timeline: build the variant with -DTR_SHADE_TIMELINE=1 and select it with TRHIP_LIB")
SEG = ["iteration overhead", "queue id", "path state", "span", "ShadeTri + instance + geometry", "albedo (ids, table, taps)", "metallic-roughness",
       "normal map", "emission, ior", "light / environment hit", "emission MIS + first-hit stores", "rng + light sample (record fetch)",
       "NEE bsdf eval + shadow record", "BSDF sample + next ray", "write-back (rmw + state stores)", "block append (3 barriers, atomics)",
       "queue + shadow stores", "calibration (two stamps)",
       "  light: rng + kind + record fetch", "  light: triangle light sample", "  light: environment sample (+ taps)", "  light: directional light sample"]
N, B = 24, 8
ctx = R.Context(0)
sc = scenes.WORKLOADS[wl](W, H)
opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)
rr = R.RtRenderer(ctx, sc, opt, (W, H), strategy=DISTRIBUTION_SHUFFLED_STRIPS, rank=world - 1, world_size=world, use_torch=False)
for _ in range(4):
    rr.reset_accumulation(); rr.render_partial(); rr.sync()
buf = (C.c_uint64 * (B * 2 * N))()
lib.trhip_debug_shade_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.trhip_debug_shade_timeline(None, 1) == 0
for _ in range(frames):
    rr.reset_accumulation(); rr.render_partial(); rr.sync()
assert lib.trhip_debug_shade_timeline(buf, 0) == 0
rr.close()
mode = os.environ.get("TRHIP_SHADE_TL_MODE", "1")
print(f"# phase timeline of k_shade, {wl} {W}x{H}, share 1/{world}, {frames} frames, 4 bounces, one frame at a time; instrument mode {mode}"
      f" ({'every stamp waits for all outstanding memory operations' if mode == '1' else 'stamps do not wait'})")
print("# clocks per wave and segment (s_memtime counts at ~2.36 GHz on this part, profiles/r5/trace_phase_timeline.txt); `taken` = share of the bounce's wave iterations in which some lane ran the segment")
print("# the `light:` rows split the light-sample segment; `rng + light sample` is then what is left of it (the early-out of a failed triangle sample, the return)")
tot_all = 0
for b in range(4):
    s = [int(buf[b * 2 * N + k]) for k in range(N)]
    c = [int(buf[b * 2 * N + N + k]) for k in range(N)]
    it = c[0]
    if not it:
        continue
    total = sum(s[k] for k in range(len(SEG)) if k != 17)
    tot_all += total
    print(f"\n## bounce {b}: {it} wave iterations over {frames} frames ({it / frames:.0f} waves per launch), {total / it:.0f} clocks per wave iteration")
    print(f"{'segment':>38} {'clocks/taken':>13} {'taken':>7} {'clocks/iter':>12} {'share':>7}")
    for k, name in enumerate(SEG):
        if not c[k]:
            continue
        print(f"{name:>38} {s[k] / c[k]:>13.0f} {c[k] / it:>7.1%} {s[k] / it:>12.0f} {(s[k] / total if k != 17 else 0):>7.1%}")
