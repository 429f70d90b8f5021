"""ctypes binding of libtrhip_comm.so (include/trhip_comm.h): the RCCL exchange for one process per GPU, and the exchange
object RtRenderer takes in place of torch.distributed's point-to-point calls.

Replaces tr::device_transfer (src/device_transfer.cc:21-347).  The library is separate from libtrhip.so (it is linked against
librccl); nothing here falls back to anything: a missing library or a failing RCCL call raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List

from .distribution import DistributionParams
from .transfer import partial_shape

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRHIP_COMM_LIB", os.path.join(_HERE, "libtrhip_comm.so"))
ID_BYTES = 128

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "trhip_comm_last_error": (C.c_char_p, []),
    "trhip_comm_unique_id": (_i, [_vp]),
    "trhip_comm_create": (_i, [_i, _i, _i, _vp, C.POINTER(_vp)]),
    "trhip_comm_destroy": (None, [_vp]),
    "trhip_comm_rank": (_i, [_vp]),
    "trhip_comm_size": (_i, [_vp]),
    "trhip_comm_get_info": (_i, [_vp, _vp]),
    "trhip_gather_partials": (_i, [_vp, _i, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz), _vp]),
    "trhip_reduce_samples": (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    "trhip_ipc_create": (_i, [_i, _i, _i, _i, _sz, _i, C.POINTER(_vp)]),
    "trhip_ipc_export": (_i, [_vp, _vp]),
    "trhip_ipc_connect": (_i, [_vp, _vp]),
    "trhip_ipc_gather_partials": (_i, [_vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz), _vp]),
    "trhip_ipc_release": (_i, [_vp, _vp]),
    "trhip_ipc_check": (_i, [_vp]),
    "trhip_ipc_destroy": (None, [_vp]),
}
IPC_EXPORT_BYTES = 256
_LIB = None


class TrhipCommError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TrhipCommError(f"{LIB_PATH} is missing: build it with `make -C tauray_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise TrhipCommError(lib().trhip_comm_last_error().decode("utf-8", "replace"))


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    _check(lib().trhip_comm_unique_id(buf))
    return buf.raw


class Comm:
    """One rank of an RCCL communicator (ncclCommInitRank; collective over all ranks that hold `uid`)."""

    def __init__(self, hip_device: int, nranks: int, rank: int, uid: bytes):
        assert len(uid) == ID_BYTES
        h = C.c_void_p()
        _check(lib().trhip_comm_create(hip_device, nranks, rank, C.create_string_buffer(uid, ID_BYTES), C.byref(h)))
        self.h, self.rank, self.nranks = h.value, rank, nranks

    def gather_partials(self, root: int, send_ptr, send_bytes: int, recv_ptrs=None, recv_bytes=None, stream=None):
        n = self.nranks
        if self.rank == root:
            ptrs = (C.c_void_p * n)(*[(p or None) for p in recv_ptrs])
            sizes = (C.c_size_t * n)(*recv_bytes)
            _check(lib().trhip_gather_partials(self.h, root, None, 0, ptrs, sizes, stream))
        else:
            _check(lib().trhip_gather_partials(self.h, root, send_ptr, send_bytes, None, None, stream))

    def info(self) -> dict:
        """trhip_comm_get_info: what RCCL says about this communicator (ranks, this rank, its HIP device, the library's version)."""
        class Info(C.Structure):
            _fields_ = [("struct_size", C.c_uint32), ("nranks", C.c_int32), ("rank", C.c_int32), ("hip_device", C.c_int32), ("rccl_version", C.c_int32)]
        i = Info()
        _check(lib().trhip_comm_get_info(self.h, C.byref(i)))
        return {"nranks": i.nranks, "rank": i.rank, "hip_device": i.hip_device, "rccl_version": i.rccl_version}

    def reduce_samples(self, root: int, send_ptr, recv_ptr, float_count: int, stream=None):
        _check(lib().trhip_reduce_samples(self.h, root, send_ptr, recv_ptr, float_count, stream))

    def close(self):
        if self.h:
            lib().trhip_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(x):
    return x.data_ptr() if hasattr(x, "data_ptr") else int(x)


class NativeExchange:
    """What RtRenderer(exchange=...) calls per frame of a pixel-sharded job: the partial frames travel to rank 0 through
    trhip_gather_partials (one grouped ncclSend / ncclRecv exchange on the default stream, which the renderer has ordered behind
    the path tracing and in front of the stitch)."""

    def __init__(self, comm: Comm):
        self.comm = comm
        self.boxes: Dict[int, tuple] = {}

    def attach(self, rank: int, ctx):
        if rank != self.comm.rank:
            raise ValueError("NativeExchange: the communicator belongs to another rank")

    def gather_to_display(self, color, dists: List[DistributionParams], rank: int, world_size: int, viewports: int, recv_buffers, ctx):
        if world_size == 1:
            return {}
        if rank != 0:
            shape = partial_shape(dists[rank], viewports)
            self.comm.gather_partials(0, _ptr(color), shape[0] * shape[1] * shape[2] * 16)
            return {}
        ptrs, sizes, out = [None] * world_size, [0] * world_size, {}
        for r in range(1, world_size):
            shape = partial_shape(dists[r], viewports)
            nbytes = shape[0] * shape[1] * shape[2] * 16
            box = self.boxes.get(r)
            if box is None or box[1] != shape:
                ctx.sync()          # a stitch of the previous shape may still read the old buffer
                box = (ctx.alloc(max(nbytes, 16)), shape)
                self.boxes[r] = box
            ptrs[r], sizes[r], out[r] = box[0].data_ptr(), nbytes, box[0]
        self.comm.gather_partials(0, None, 0, ptrs, sizes)
        return out


class Ipc:
    """One rank's end of the copy-engine exchange (trhip_ipc_*, include/trhip_comm.h).  `allgather(blob) -> [blob of rank 0, ...]`
    is the caller's transport for the set-up (torch.distributed.all_gather_object, a pipe, files)."""

    def __init__(self, hip_device: int, nranks: int, rank: int, slot_bytes: int, slots: int, allgather, root: int = 0):
        h = C.c_void_p()
        _check(lib().trhip_ipc_create(hip_device, nranks, rank, root, slot_bytes, slots, C.byref(h)))
        self.h, self.rank, self.nranks, self.root, self.slot_bytes, self.slots = h.value, rank, nranks, root, slot_bytes, slots
        blob = C.create_string_buffer(IPC_EXPORT_BYTES)
        _check(lib().trhip_ipc_export(self.h, blob))
        blobs = allgather(blob.raw)
        assert len(blobs) == nranks and all(len(b) == IPC_EXPORT_BYTES for b in blobs)
        _check(lib().trhip_ipc_connect(self.h, C.create_string_buffer(b"".join(blobs), IPC_EXPORT_BYTES * nranks)))

    def send(self, send_ptr, send_bytes: int, stream=None):
        _check(lib().trhip_ipc_gather_partials(self.h, send_ptr, send_bytes, None, None, stream))

    def receive(self, recv_bytes, stream=None):
        """Root: {peer: device pointer of its partial frame}; the pointers stay valid until release()."""
        ptrs = (C.c_void_p * self.nranks)()
        sizes = (C.c_size_t * self.nranks)(*recv_bytes)
        _check(lib().trhip_ipc_gather_partials(self.h, None, 0, ptrs, sizes, stream))
        return {r: ptrs[r] for r in range(self.nranks) if ptrs[r]}

    def release(self, stream=None):
        _check(lib().trhip_ipc_release(self.h, stream))

    def check(self):
        """Raises if a device-side wait for a peer has given up (trhip_ipc_check): call after synchronising the stream of a frame that is
        about to be used - the last frame of a job is followed by no other call into the exchange."""
        _check(lib().trhip_ipc_check(self.h))

    def close(self):
        if self.h:
            lib().trhip_ipc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _ArenaView:
    """A partial frame inside the display rank's receive arena, as the stitch takes it."""

    def __init__(self, ptr):
        self.ptr = ptr

    def data_ptr(self):
        return self.ptr


class IpcExchange:
    """RtRenderer(exchange=...) over the copy engines: the partial frames of a pixel-sharded job are written by their ranks into the
    display rank's IPC-mapped receive arena (hipMemcpyAsync - a DMA over the sender's xGMI link, no kernel on either device), tags
    order them (trhip_ipc_*).  The arena slot of a frame is released when the next frame is gathered: by then the renderer has
    enqueued the stitch that reads it, in front of the release on the same stream."""

    def __init__(self, ipc: Ipc):
        self.ipc = ipc
        self.pending_release = False

    def attach(self, rank: int, ctx):
        if rank != self.ipc.rank:
            raise ValueError("IpcExchange: the exchange end belongs to another rank")

    def check(self):
        """RtRenderer.sync() / download() call this behind their synchronisation: a frame whose wait for a peer gave up is an error, not an image."""
        self.ipc.check()

    def gather_to_display(self, color, dists: List[DistributionParams], rank: int, world_size: int, viewports: int, recv_buffers, ctx):
        if world_size == 1:
            return {}
        if rank != 0:
            shape = partial_shape(dists[rank], viewports)
            self.ipc.send(_ptr(color), shape[0] * shape[1] * shape[2] * 16)
            return {}
        if self.pending_release:
            self.ipc.release()      # the previous frame's stitch sits in front of this on the stream
        sizes = [0] * world_size
        for r in range(1, world_size):
            shape = partial_shape(dists[r], viewports)
            sizes[r] = shape[0] * shape[1] * shape[2] * 16
        ptrs = self.ipc.receive(sizes)
        self.pending_release = True
        return {r: _ArenaView(p) for r, p in ptrs.items()}
