"""What needs two or more real devices: the exchanges between them.  Collected everywhere, skipped on a box with fewer than two GPUs (the
development pool has one per box), run by the driver's `pytest -m gpu` the first time a multi-GPU node is leased - so that lease proves
the transports correct and does not only print a rate (tools/first_multi_gpu_run.sh runs these and then the A/B matrix of DESIGN.md section 6).

Every other multi-GPU test of the tree moves its bytes without leaving a device: fake devices inside one process (tests/test_gpu_parity.py,
test_cpp_host.py), several processes sharing device 0 (tests/test_comm.py), gloo on the CPU (tests/test_multi_rank_gloo.py).  What those
cannot reach, and these do:
  * RCCL's ncclSend / ncclRecv between ranks (trhip_gather_partials; RCCL refuses two ranks on one device) and ncclReduce
    (trhip_reduce_samples) - replaces src/device_transfer.cc:140-290 and the per-frame copies of src/rt_renderer.cc:84-133;
  * trhip_ipc_*: hipIpcOpenMemHandle of another *device's* allocation (peer access enabled lazily), a DMA over xGMI into it, a
    system-scope poll that sees a tag a peer's copy engine wrote - with the fine-grained arena and with TRHIP_IPC_COARSE=1;
  * tr::rt_renderer on real devices: hipMemcpyPeerAsync + events waited for across devices (trhip_copy_peer, trhip_stream_wait_peer);
  * bench.py --gpus N as the driver launches it, display frame against N = 1.
The reference frame of every test is one process on device 0."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, device_count

N_DEV = device_count()
# TRHIP_TEST_MULTI_DEVICE_REHEARSAL=1 on a one-GPU box: the same jobs with every rank on device 0 - no byte crosses a link, and the RCCL
# jobs cannot run at all (RCCL refuses two ranks on one device), but the command lines, scripts and comparisons of this file are exercised
REHEARSAL = N_DEV == 1 and os.environ.get("TRHIP_TEST_MULTI_DEVICE_REHEARSAL") == "1"
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(N_DEV < 2 and not REHEARSAL, reason=f"needs two or more GPUs on one node (hipGetDeviceCount() = {N_DEV}): the RCCL / IPC / "
                                                                     "peer-copy exchanges between real devices")]
needs_rccl_ranks = pytest.mark.skipif(REHEARSAL, reason="rehearsal on one device: RCCL refuses two ranks on one device")


def _dev(rank):
    return 0 if REHEARSAL else rank


CLI = os.path.join(ROOT, "tauray_amd", "tauray_hip")
WORLDS = sorted({2, min(N_DEV, 4), N_DEV} - {0, 1}) if N_DEV >= 2 else [2, 3]
ENV = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.fixture(scope="module")
def scene_dump(tmp_path_factory, test_glb_128):
    """test.glb as the host layer's scene dump (what tests/test_cpp_host.py feeds the command line)."""
    from tauray_amd.scene_io import write_scene_dump
    p = str(tmp_path_factory.mktemp("md") / "test.trsc")
    write_scene_dump(test_glb_128, p)
    return p


def _frames(prefix, n):
    return [np.fromfile(f"{prefix}{f}.raw", dtype=np.float32) for f in range(n)]


def _reference(tmp_path, common, n, tag="ref"):
    prefix = str(tmp_path / tag)
    subprocess.check_call([CLI] + common + [f"--headless={prefix}", "--devices=0"], env=ENV)
    return _frames(prefix, n)


# --------------------------------------------------------------------------------------------------------------------------------------
# (iii) one process, several devices: the reference's own organisation (tr::rt_renderer, include/tauray_hip.hh)
@pytest.mark.parametrize("strategy", ["scanline", "shuffled-strips"])
@pytest.mark.parametrize("slots", [1, 4])
def test_one_process_drives_real_devices(tmp_path, scene_dump, strategy, slots):
    """tauray_hip --devices=0,1,..: every device traces its share, hipMemcpyPeerAsync carries it to the display device behind an event of
    the tracing stream, the display device stitches behind events of the copies; four frames in flight.  Files equal one device's."""
    W, H, F = 192, 136, 6
    common = [scene_dump, f"--width={W}", f"--height={H}", "--max-ray-depth=4", "--filetype=raw", f"--frames={F}"]
    ref = _reference(tmp_path, common, F)
    for world in WORLDS:
        prefix = str(tmp_path / f"dev{world}")
        devs = ",".join(str(_dev(d)) for d in range(world))
        r = subprocess.run([CLI] + common + [f"--headless={prefix}", f"--devices={devs}", f"--distribution-strategy={strategy}", f"--frames-in-flight={slots}"],
                           capture_output=True, text=True, timeout=600, env=ENV)
        assert r.returncode == 0, r.stderr[-3000:]
        got = _frames(prefix, F)
        assert all(np.array_equal(g, e) for g, e in zip(got, ref)), (world, strategy, slots)


# --------------------------------------------------------------------------------------------------------------------------------------
# (i) + (ii) one process per GPU through the command line: RCCL gather and the copy-engine exchange
def _process_job(tmp_path, common, world, tag, extra, env=None, frames=5):
    prefix, idf = str(tmp_path / tag), str(tmp_path / (tag + ".id"))
    procs = [subprocess.Popen([CLI] + common + [f"--headless={prefix}", f"--process-count={world}", f"--process-rank={r}", f"--device={_dev(r)}", f"--comm-id={idf}",
                                                f"--comm-nonce={os.getpid() * 131 + len(tag)}"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env or ENV)
             for r in range(world)]
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, (tag, se[-3000:])
    return _frames(prefix, frames)


@needs_rccl_ranks
@pytest.mark.parametrize("strategy", ["scanline", "shuffled-strips"])
def test_ranks_gather_over_rccl(tmp_path, scene_dump, strategy):
    """trhip_gather_partials with 2 ... N real ranks, one per device: grouped ncclSend / ncclRecv into the display rank, stitch, tonemap;
    balanced strips (unequal shares: --device-workloads) and scanlines; two frames in flight.  Display frames bit-equal to one rank's."""
    W, H, F = 192, 136, 5
    common = [scene_dump, f"--width={W}", f"--height={H}", "--max-ray-depth=4", "--filetype=raw", f"--frames={F}"]
    ref = _reference(tmp_path, common, F)
    for world in WORLDS:
        extra = [f"--distribution-strategy={strategy}"]
        if strategy == "shuffled-strips":
            shares = [1.0 + 0.5 * (r % 3) for r in range(world)]
            extra.append("--device-workloads=" + ",".join(f"{s / sum(shares):.4f}" for s in shares))
        for slots in (1, 2):
            got = _process_job(tmp_path, common, world, f"rccl{world}_{slots}", extra + [f"--frames-in-flight={slots}"], frames=F)
            assert all(np.array_equal(g, e) for g, e in zip(got, ref)), (world, strategy, slots)


@pytest.mark.parametrize("coarse", [False, True])
def test_ranks_exchange_through_the_copy_engines_across_devices(tmp_path, scene_dump, coarse):
    """trhip_ipc_* between devices: the display rank's arena and tags are opened by the other devices' processes (peer access enabled on
    first use), partial frames arrive by hipMemcpyAsync over xGMI, the tags behind them; fine-grained arena (default) and plain hipMalloc."""
    W, H, F = 192, 136, 6
    common = [scene_dump, f"--width={W}", f"--height={H}", "--max-ray-depth=4", "--filetype=raw", f"--frames={F}"]
    ref = _reference(tmp_path, common, F)
    env = dict(ENV, TRHIP_IPC_COARSE="1") if coarse else ENV
    for world in WORLDS:
        for slots, strategy in ((1, "scanline"), (2, "shuffled-strips")):
            got = _process_job(tmp_path, common, world, f"ipc{world}_{slots}_{int(coarse)}", ["--exchange=ipc", f"--frames-in-flight={slots}", f"--distribution-strategy={strategy}"],
                               env=env, frames=F)
            assert all(np.array_equal(g, e) for g, e in zip(got, ref)), (world, slots, strategy, coarse)


def test_a_dead_peer_on_another_device_is_an_error(tmp_path):
    """The display rank's device-side wait for a peer on another device gives up and the next call says so (tests/test_comm.py has the
    one-device form)."""
    import test_comm
    script = tmp_path / "rank.py"
    script.write_text(test_comm._IPC_SILENT_PEER)
    work = tmp_path / "work"
    work.mkdir()
    env = dict(ENV, TRHIP_IPC_TIMEOUT_MS="300", **({} if REHEARSAL else {"TRHIP_TEST_DEVICE_PER_RANK": "1"}))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(work)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("SECOND CALL")][0]
    assert line.startswith("SECOND CALL FAILED AFTER") and "gave up" in line, line


def test_python_ranks_exchange_across_devices(tmp_path):
    """The Python mirror's IpcExchange with a rank per device (tests/test_comm.py's job with TRHIP_TEST_DEVICE_PER_RANK)."""
    import test_comm
    from tauray_amd import renderer as R, scenes
    frames, world, slots = 6, WORLDS[-1] if WORLDS[-1] <= 4 else 4, 2
    script = tmp_path / "rank.py"
    script.write_text(test_comm._IPC_RANK)
    work = tmp_path / "work"
    work.mkdir()
    env = dict(ENV, **({} if REHEARSAL else {"TRHIP_TEST_DEVICE_PER_RANK": "1"}))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(work), str(frames), str(slots)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
    W, H = 256, 192
    scene = scenes.test_glb(W, H)
    ctx = R.Context(0)
    rr = R.RtRenderer(ctx, scene, R.options_for_scene(scene, max_bounces=3), (W, H))
    for f in range(frames):
        rr.render()
    rr.sync()
    ref = rr.download("display").copy()
    rr.close()
    assert np.array_equal(np.load(work / "last.npy"), ref)


# --------------------------------------------------------------------------------------------------------------------------------------
# (iv) sample shards: ncclReduce(sum) of the partial sums
_REDUCE_RANK = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
rank, world, workdir = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
import numpy as np
from tauray_amd import comm, renderer as R
idf = os.path.join(workdir, "id")
if rank == 0:
    open(idf + ".tmp", "wb").write(comm.unique_id()); os.rename(idf + ".tmp", idf)
t0 = time.time()
while not os.path.exists(idf):
    assert time.time() - t0 < 120
    time.sleep(0.01)
uid = open(idf, "rb").read()
ctx = R.Context(rank)
c = comm.Comm(rank, world, rank, uid)
n = 1920 * 1080 * 4
rng = np.random.default_rng(100 + rank)
mine = rng.random(n, dtype=np.float32)
src, dst = ctx.alloc(n * 4).upload(mine), ctx.alloc(n * 4).zero()
st = ctx.create_stream()
for _ in range(3):
    c.reduce_samples(0, src.data_ptr(), dst.data_ptr(), n, st)
ctx.sync(st)
if rank == 0:
    np.save(os.path.join(workdir, "sum.npy"), dst.download((n,)))
c.close()
"""


@needs_rccl_ranks
def test_sample_sums_reduce_over_rccl(tmp_path):
    """trhip_reduce_samples: W x H x 4 floats of every rank summed on the display rank (config 3's sample shards); against numpy in the
    order a ring may take (tolerance of a float sum of N terms, not bits)."""
    world = WORLDS[-1]
    script = tmp_path / "rank.py"
    script.write_text(_REDUCE_RANK)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(tmp_path)], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
    n = 1920 * 1080 * 4
    want = np.zeros(n, dtype=np.float64)
    for r in range(world):
        want += np.random.default_rng(100 + r).random(n, dtype=np.float32)
    got = np.load(tmp_path / "sum.npy")
    assert np.abs(got - want).max() < 1e-6 * world * world


# --------------------------------------------------------------------------------------------------------------------------------------
# (v) the bench as the driver launches it
@pytest.mark.parametrize("exchange", ["native", "ipc"] if not REHEARSAL else ["torch", "ipc"])
def test_bench_command_line_renders_the_one_gpu_display_frame(tmp_path, exchange):
    """python -m torch.distributed.run --nproc-per-node N bench.py --gpus N --steps 20 --save-display: the JSON line parses, the display
    frame of frame 0 equals the N = 1 frame (strips with balanced shares), for RCCL's gather and for the copy-engine exchange."""
    import json
    one = str(tmp_path / "one.npy")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--prewarm", "0", "--no-cpu-baseline", "--no-roofline",
                        "--sustained-frames", "0", "--save-display", one], capture_output=True, text=True, timeout=900, env=ENV, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    for world in WORLDS:
        out = str(tmp_path / f"n{world}_{exchange}.npy")
        port = 29500 + (os.getpid() + world * 7 + len(exchange)) % 400
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "3", "--prewarm", "10", "--exchange", exchange,
                            "--save-display", out] + (["--dist-backend", "gloo", "--one-device"] if REHEARSAL else []),
                           capture_output=True, text=True, timeout=1800, env=ENV, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == world and line["steps"] == 20 and line["value"] > 0 and line["scaling"] == "strong"
        assert np.array_equal(np.load(out), np.load(one)), (world, exchange)
