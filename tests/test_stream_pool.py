"""Streams by hardware pipe (csrc/stream_pool.hip, profiles/r5/stream_pipes.txt).  The reference leaves queue placement to the Vulkan
driver (one compute queue per device, src/context.cc); here a lone frame runs as four lanes on four streams, and lanes whose queues
share a pipe run one after the other - so the library classifies every stream once and spreads lanes and frame slots over the pipes.
These tests pin the two things that makes true: the classes are found, and a renderer's speed does not depend on which renderers the
process had before it."""
import os
import time

import pytest



@pytest.fixture(scope="module")
def R():
    from tauray_amd import renderer
    return renderer


@pytest.fixture(scope="module")
def ctx(R):
    yield R.Context(0)


@pytest.mark.gpu
def test_pool_streams_are_spread_over_the_pipes(R, ctx):
    null_class = ctx.stream_pipe_class(None)
    ss = [ctx.create_stream() for _ in range(4)]
    try:
        classes = [ctx.stream_pipe_class(s) for s in ss]
        assert null_class >= 0 and all(c >= 0 for c in classes)
        known = len(set(classes + [null_class]))
        assert known >= 2, "the pipe experiment found one pipe only"      # gfx950 under ROCm 7.2: four
        # four slots of a renderer: on as many pipes as there are (no two on one pipe while another pipe has none); with the runtime's
        # default of four hardware queues the null stream holds one of them and the fourth stream has to share
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        assert len(set(classes)) == min(4 if queues >= 8 else 3, known), (null_class, classes)
        assert ctx.stream_pipe_class(ss[0]) == classes[0]                 # a stream keeps its class
    finally:
        for s in ss:
            ctx.destroy_stream(s)
    again = [ctx.create_stream() for _ in range(4)]                       # the same streams come back: nothing is destroyed
    try:
        assert sorted(again) == sorted(ss)
    finally:
        for s in again:
            ctx.destroy_stream(s)


@pytest.mark.gpu
def test_a_renderer_is_as_fast_after_other_renderers_as_before(R, ctx):
    """Before the pool a 1/8 strip took 0.79 ms in a fresh process and 1.00 ms once a four-slot renderer had existed (its lanes' new
    streams landed on one pipe); a quarter slower is what this bound would catch, run-to-run spread is 2 %."""
    from tauray_amd import scenes
    from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS
    W, H = 1920, 1080
    sc = scenes.WORKLOADS["sponza_class"](W, H)
    opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)

    def strip_ms(lanes):
        rr = R.RtRenderer(ctx, sc, opt, (W, H), strategy=DISTRIBUTION_SHUFFLED_STRIPS, rank=7, world_size=8, use_torch=False)
        rr.slots[0].pt.set_lanes(lanes)
        best = 1e9
        for rep in range(3):
            for i in range(8 + 48):
                if i == 8:
                    t0 = time.perf_counter()
                rr.reset_accumulation(); rr.render_partial(); rr.sync()
            best = min(best, (time.perf_counter() - t0) / 48 * 1e3)
        rr.close()
        return best

    # a whole frame runs as four lanes, each on a pipe of its own, whatever streams exist already
    whole = R.RtRenderer(ctx, sc, opt, (W, H), use_torch=False)
    whole.render_partial(); whole.sync()
    lanes, pipes = whole.slots[0].pt.lane_pipes()
    whole.close()
    probe = [ctx.create_stream() for _ in range(8)]
    known = len({ctx.stream_pipe_class(s) for s in probe} | {ctx.stream_pipe_class(None)})      # pipes this process can reach: four on gfx950
    for s in probe:
        ctx.destroy_stream(s)
    assert lanes == 4 and -1 not in pipes and len(set(pipes)) == min(4, known), (lanes, pipes, known)
    before = {l: strip_ms(l) for l in (2, 4)}
    held = [ctx.create_stream() for _ in range(3)]            # an application's own streams, idle
    slots = R.RtRenderer(ctx, sc, opt, (W, H), use_torch=False, frames_in_flight=4, frames_per_launch=2)
    for _ in range(4):
        slots.render_partial()
    slots.sync(); slots.close()
    after = {l: strip_ms(l) for l in (2, 4)}
    for s in held:
        ctx.destroy_stream(s)
    # Wall-clock bounds inside a correctness suite: wide by default (a noisy neighbour on the lease must not turn a parity round red -
    # what the pool prevents is a factor of 1.27), the measured margins with TRHIP_PERF_TESTS=1 (profiles/r5/stream_pipes.txt)
    strict = os.environ.get("TRHIP_PERF_TESTS") == "1"
    for l in (2, 4):
        assert after[l] < before[l] * (1.12 if strict else 1.5), (before, after)
    if strict:
        assert before[4] < before[2] * 1.05, before           # four lanes on four pipes are no slower than two (0.61 against 0.70 ms)


@pytest.mark.gpu
def test_two_slots_run_two_lanes_each_and_render_the_same_frames(R, ctx):
    """A renderer with the reference's two frames in flight (MAX_FRAMES_IN_FLIGHT, src/context.hh:26) runs two lanes per slot, the four
    on four pipes (above 50 k paths); three slots run one lane each.  Frames are the one-frame-at-a-time renderer's, bit for bit."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from tauray_amd.gltf import load_glb
    W, H = 384, 256
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    opt = R.options_for_scene(scene, max_bounces=3)
    serial = R.RtRenderer(ctx, scene, opt, (W, H), use_torch=False)
    want = []
    for _ in range(5):
        serial.render()
        want.append(serial.download("color"))
    serial.close()
    for F, lanes_expected in ((2, 2), (3, 1)):
        rr = R.RtRenderer(ctx, scene, opt, (W, H), use_torch=False, frames_in_flight=F)
        for i in range(5):
            rr.render()
        rr.sync()
        pipes = []
        for slot in rr.slots:
            lanes, p = slot.pt.lane_pipes()
            assert lanes == lanes_expected, (F, lanes)
            pipes += p
        if -1 not in pipes and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 8:
            assert len(set(pipes)) == min(len(pipes), 4), (F, pipes)      # no two lanes of the renderer on one pipe
        for back in range(F):
            i = 4 - back
            assert np.array_equal(rr.slots[i % F].color.download((1, H, W, 4)), want[i]), f"F={F}: frame {i}"
        rr.close()


CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from tauray_amd import renderer as R
from tauray_amd.gltf import load_glb
W, H = 384, 256
scene = load_glb(os.path.join(sys.argv[1], "tests", "golden", "test.glb"), W, H)
ctx = R.Context(0)
info = ctx.info()
rr = R.RtRenderer(ctx, scene, R.options_for_scene(scene, max_bounces=3), (W, H), use_torch=False)
rr.slots[0].pt.set_lanes(4)
rr.render(); rr.sync()
lanes, pipes = rr.slots[0].pt.lane_pipes()
np.save(sys.argv[2], rr.download("color"))
rr.close()
print(json.dumps({"info": info, "lanes": lanes, "pipes": pipes}))
"""


@pytest.mark.gpu
def test_a_process_without_eight_hardware_queues_is_told_and_renders_the_same_bits(tmp_path):
    """include/trhip.h "process requirements": GPU_MAX_HW_QUEUES >= 8 before the first HIP call.  A process that does not meet it (the
    runtime's default of four queues) renders the same frames, slower; trhip_device_get_info reports how many pipes it reaches, and with
    TRHIP_DEBUG=1 the library says so on stderr once - when fewer than four pipes are reachable, or when two lanes of a frame share one."""
    import json
    import subprocess
    import sys
    import numpy as np
    from conftest import ROOT
    out = {}
    for queues in ("8", "4"):
        npy = str(tmp_path / f"frame_{queues}.npy")
        env = dict(os.environ, GPU_MAX_HW_QUEUES=queues, TRHIP_DEBUG="1")
        env.pop("TRHIP_PIPE_CLASSES", None)
        p = subprocess.run([sys.executable, "-c", CHILD, ROOT, npy], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        out[queues] = (rec, np.load(npy), p.stderr)
    (r8, f8, e8), (r4, f4, e4) = out["8"], out["4"]
    assert np.array_equal(f8, f4), "frames depend on the number of hardware queues"
    assert r8["info"]["hw_queues_env"] == 8 and r4["info"]["hw_queues_env"] == 4
    assert r8["info"]["pipe_classes"] >= 2 and r8["info"]["name"].startswith("gfx")
    assert r8["lanes"] == 4 and r4["lanes"] == 4
    for rec, err in ((r8, e8), (r4, e4)):
        known = [c for c in rec["pipes"] if c >= 0]
        shared = len(set(known)) < len(known)
        told = "hardware pipe" in err
        if rec["info"]["pipe_classes"] < 4 or shared:
            assert told, (rec, err[-500:])
    if r8["info"]["pipe_classes"] >= 4:      # gfx950 under ROCm 7.2: the requirement met means four lanes on four pipes, and no complaint
        assert len(set(r8["pipes"])) == 4 and -1 not in r8["pipes"] and "share a hardware pipe" not in e8, (r8, e8[-500:])


@pytest.mark.gpu
def test_pinned_pipe_classes_skip_the_experiment(tmp_path):
    """TRHIP_PIPE_CLASSES=0,1,2,3: the library's streams take these classes in creation order, nothing is probed, foreign streams are -1."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, sys.argv[1])\n"
            "from tauray_amd import renderer as R\n"
            "ctx = R.Context(0)\n"
            "ss = [ctx.create_stream() for _ in range(6)]\n"
            "print([ctx.stream_pipe_class(s) for s in ss], ctx.stream_pipe_class(None), ctx.info()['pipe_classes'])\n")
    p = subprocess.run([sys.executable, "-c", code, ROOT], env=dict(os.environ, TRHIP_PIPE_CLASSES="0,1,2,3"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.strip().splitlines()[-1] == "[0, 1, 2, 3, 0, 1] -1 4", p.stdout
