"""libtrhip_comm.so (include/trhip_comm.h): the RCCL exchange for one process per GPU.  Without a GPU: the library loads (it is
linked against librccl) and exports every declared symbol.  On the one-GPU box: everything a communicator of one rank can show -
creation, the gather as a no-op, the reduce as a copy on the caller's stream - and the bench's multi-rank command line with
the ranks sharing the device over gloo (tests/test_gpu_parity.py covers the stitch the received frames go into)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_comm_library_exports_every_declared_symbol():
    from tauray_amd import comm
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "trhip_comm.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(trhip_[a-z_0-9]+)\s*\(", text)))
    L = comm.lib()
    assert len(names) == 16 and sorted(comm.SYMBOLS) == names
    for n in names:
        assert hasattr(L, n), f"libtrhip_comm.so does not export {n}"
    # linked against RCCL, not against the path-tracing library
    import subprocess
    deps = subprocess.run(["ldd", comm.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in deps and "libtrhip.so" not in deps


@pytest.mark.gpu
def test_one_rank_communicator():
    from tauray_amd import comm
    from tauray_amd import renderer as R
    ctx = R.Context(0)
    uid = comm.unique_id()
    assert len(uid) == comm.ID_BYTES
    c = comm.Comm(0, 1, 0, uid)
    assert c.rank == 0 and c.nranks == 1 and comm.lib().trhip_comm_size(c.h) == 1
    info = c.info()      # asked of RCCL, not echoed (trhip_comm_get_info): what the N > 1 bench line prints
    assert info["nranks"] == 1 and info["rank"] == 0 and info["hip_device"] == 0 and info["rccl_version"] > 20000
    n = 1920 * 136 * 4
    src = ctx.alloc(n * 4).upload(np.arange(n, dtype=np.float32))
    dst = ctx.alloc(n * 4).zero()
    st = ctx.create_stream()
    c.reduce_samples(0, src.data_ptr(), dst.data_ptr(), n, st)        # a sum over one rank: the rank's own samples
    ctx.sync(st)
    assert np.array_equal(dst.download((n,)), np.arange(n, dtype=np.float32))
    c.reduce_samples(0, src.data_ptr(), src.data_ptr(), n, st)        # in place
    c.gather_partials(0, None, 0, [None], [0], st)                    # nothing to receive
    ctx.sync(st)
    assert np.array_equal(src.download((n,)), np.arange(n, dtype=np.float32))
    with pytest.raises(comm.TrhipCommError):
        c.gather_partials(3, None, 0, [None], [0])                    # root out of range
    ex = comm.NativeExchange(c)
    assert ex.gather_to_display(src, [], 0, 1, 1, {}, ctx) == {}
    c.close()
    ctx.destroy_stream(st)


_IPC_RANK = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
rank, world, workdir, frames, slots = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
import numpy as np
from tauray_amd import comm, renderer as R, scenes
from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS, get_distribution_target_max_size


def allgather(blob):      # the caller's transport for the set-up: files in a directory both processes see
    open(os.path.join(workdir, f"blob{rank}.tmp"), "wb").write(blob)
    os.rename(os.path.join(workdir, f"blob{rank}.tmp"), os.path.join(workdir, f"blob{rank}"))
    out, t0 = [], time.time()
    for r in range(world):
        path = os.path.join(workdir, f"blob{r}")
        while not os.path.exists(path):
            assert time.time() - t0 < 120, "the other rank never showed up"
            time.sleep(0.01)
        out.append(open(path, "rb").read())
    return out


W, H = 256, 192
scene = scenes.test_glb(W, H)
dev = rank if os.environ.get("TRHIP_TEST_DEVICE_PER_RANK") else 0      # every rank on device 0: two processes, one GPU (tests/test_multi_device.py: one each)
ctx = R.Context(dev)
opt = R.options_for_scene(scene, max_bounces=3)
ipc = comm.Ipc(dev, world, rank, W * H * 16, slots, allgather)
rr = R.RtRenderer(ctx, scene, opt, (W, H), strategy=DISTRIBUTION_SHUFFLED_STRIPS, rank=rank, world_size=world, exchange=comm.IpcExchange(ipc), frames_in_flight=slots)
out = []
for f in range(frames):
    rr.render()
    if rank == 0 and (f % slots == slots - 1 or f == frames - 1):      # the display rank looks at its frames now and then; the others never wait
        rr.sync()
    if rank == 0:
        out.append(rr.download("display").copy()) if slots == 1 else None
rr.sync()
if rank == 0:
    np.save(os.path.join(workdir, "last.npy"), rr.download("display"))
    if out:
        np.save(os.path.join(workdir, "all.npy"), np.stack(out))
open(os.path.join(workdir, f"done{rank}"), "w").write("ok")
t0 = time.time()
while not all(os.path.exists(os.path.join(workdir, f"done{r}")) for r in range(world)):      # nobody unmaps while the other still copies
    assert time.time() - t0 < 120
    time.sleep(0.01)
rr.close()
ipc.close()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world,slots", [(2, 1), (3, 2)])
def test_processes_exchange_frames_through_the_copy_engines(tmp_path, world, slots):
    """trhip_ipc_* (include/trhip_comm.h) with real processes: `world` processes share the one GPU, each renders its shuffled-strip share
    of every frame, the partial frames travel into the display process's IPC-mapped arena by hipMemcpyAsync, tags order them, the display
    process stitches and tonemaps.  Bytes move between address spaces; every display frame is the single-process frame bit for bit."""
    import subprocess
    import sys
    from tauray_amd import renderer as R, scenes
    frames = 6
    script = tmp_path / "rank.py"
    script.write_text(_IPC_RANK)
    work = tmp_path / "work"
    work.mkdir()
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(work), str(frames), str(slots)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    # the same frames from one process
    W, H = 256, 192
    scene = scenes.test_glb(W, H)
    ctx = R.Context(0)
    rr = R.RtRenderer(ctx, scene, R.options_for_scene(scene, max_bounces=3), (W, H))
    ref = []
    for f in range(frames):
        rr.render()
        rr.sync()
        ref.append(rr.download("display").copy())
    rr.close()
    last = np.load(work / "last.npy")
    assert np.isfinite(last).all() and last[..., :3].mean() > 1e-3 and np.array_equal(last, ref[-1])
    if slots == 1:
        got = np.load(work / "all.npy")
        assert all(np.array_equal(got[f], ref[f]) for f in range(frames))


_IPC_SILENT_PEER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
rank, workdir = int(sys.argv[2]), sys.argv[3]
from tauray_amd import comm, renderer as R


def allgather(blob):
    open(os.path.join(workdir, f"blob{rank}.tmp"), "wb").write(blob)
    os.rename(os.path.join(workdir, f"blob{rank}.tmp"), os.path.join(workdir, f"blob{rank}"))
    out, t0 = [], time.time()
    for r in range(2):
        path = os.path.join(workdir, f"blob{r}")
        while not os.path.exists(path):
            assert time.time() - t0 < 120
            time.sleep(0.01)
        out.append(open(path, "rb").read())
    return out


dev = rank if os.environ.get("TRHIP_TEST_DEVICE_PER_RANK") else 0
ctx = R.Context(dev)
ipc = comm.Ipc(dev, 2, rank, 4096, 1, allgather)
if rank == 0:
    ipc.receive([0, 1024])      # the display rank waits for rank 1's bytes, which never come
    t = time.time()
    ctx.sync()
    waited = time.time() - t
    try:
        ipc.receive([0, 1024])
        print("SECOND CALL PASSED", waited)
    except comm.TrhipCommError as e:
        print("SECOND CALL FAILED AFTER", waited, str(e))
open(os.path.join(workdir, f"done{rank}"), "w").write("ok")
t0 = time.time()
while not all(os.path.exists(os.path.join(workdir, f"done{r}")) for r in range(2)):
    assert time.time() - t0 < 120
    time.sleep(0.01)
ipc.close()
"""


@pytest.mark.gpu
def test_a_peer_that_stops_sending_is_an_error_not_a_stale_frame(tmp_path):
    """The display rank's wait for a partial frame gives up after a while (10 s; TRHIP_IPC_TIMEOUT_MS here) so that a dead peer cannot hang a
    device for good - and the next call into the exchange says so: the frame behind that wait holds whatever was in the arena."""
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text(_IPC_SILENT_PEER)
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, TRHIP_IPC_TIMEOUT_MS="300")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(work)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("SECOND CALL")][0]
    assert line.startswith("SECOND CALL FAILED AFTER") and "gave up" in line, line
    assert 0.25 < float(line.split()[4]) < 5.0, line


_PROGRAM_RANK = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
rank, world, workdir = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
from tauray_amd import comm, renderer as R, scenes
from tauray_amd.distribution import DISTRIBUTION_SHUFFLED_STRIPS


def allgather(blob, tag=[0]):      # files in a directory both processes see; one round per call
    tag[0] += 1
    name = lambda r: os.path.join(workdir, f"blob{tag[0]}_{r}")
    open(name(rank) + ".tmp", "wb").write(blob)
    os.rename(name(rank) + ".tmp", name(rank))
    out, t0 = [], time.time()
    for r in range(world):
        while not os.path.exists(name(r)):
            assert time.time() - t0 < 120, "the other rank never showed up"
            time.sleep(0.01)
        out.append(open(name(r), "rb").read())
    return out


W, H = 96, 64
scene = scenes.test_glb(W, H)
ctx = R.Context(0)
opt = R.options_for_scene(scene, max_bounces=2, sampler=2)      # not the command-line set: a program compiled for the set, or the general kernels
ipc = comm.Ipc(0, world, rank, W * H * 16, 1, allgather)
rr = R.RtRenderer(ctx, scene, opt, (W, H), strategy=DISTRIBUTION_SHUFFLED_STRIPS, rank=rank, world_size=world, exchange=comm.IpcExchange(ipc))
try:
    p = rr.check_same_program(allgather)
    print("SAME", p["kind"], "%016x" % p["identity"])
except RuntimeError as e:
    print("DIFFERENT", str(e).replace("\n", " "))
open(os.path.join(workdir, f"done{rank}"), "w").write("ok")
t0 = time.time()
while not all(os.path.exists(os.path.join(workdir, f"done{r}")) for r in range(world)):
    assert time.time() - t0 < 120
    time.sleep(0.01)
rr.close()
ipc.close()
"""


@pytest.mark.gpu
def test_ranks_compare_their_shading_programs_before_the_first_frame(tmp_path):
    """RtRenderer.check_same_program (trhip_pt_get_program): two ranks with the same library, cache and environment agree; with one rank
    under TRHIP_SPECIALIZE=0 (what a failed run-time compilation amounts to: that rank would render its strips with the general kernels)
    both ranks refuse, naming the two programs."""
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text(_PROGRAM_RANK)
    lines = {}
    for case, env1 in (("same", {}), ("one_general", {"TRHIP_SPECIALIZE": "0"})):
        work = tmp_path / case
        work.mkdir()
        procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", str(work)], env=dict(os.environ, **(env1 if r == 1 else {})),
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
        outs = [p.communicate(timeout=600) for p in procs]
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, se[-3000:]
        lines[case] = [[l for l in so.splitlines() if l.startswith(("SAME", "DIFFERENT"))][0] for so, _ in outs]
    assert all(l.startswith("SAME compiled") for l in lines["same"]) and lines["same"][0] == lines["same"][1], lines["same"]
    assert all(l.startswith("DIFFERENT") and "general kernels" in l and "compiled" in l for l in lines["one_general"]), lines["one_general"]
