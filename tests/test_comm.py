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
    assert len(names) == 8 and sorted(comm.SYMBOLS) == names
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
