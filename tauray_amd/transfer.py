"""Inter-device frame transfer: the RCCL replacement of the reference's device_transfer
(src/device_transfer.cc:21-347: GPU -> pinned host -> GPU copies with exported semaphores).

One process per GPU.  Non-display ranks post their partial frame, the display rank (0) receives every
partial straight into device memory; all point-to-point operations of a frame go into one
`batch_isend_irecv` (= one ncclGroupStart/End), so each peer's slab travels over its own xGMI link.
Works on any torch.distributed backend ("nccl" = RCCL on GPUs, "gloo" on CPU tensors in the tests).
"""
from __future__ import annotations

from typing import Dict, List

from .distribution import DistributionParams, get_distribution_target_size


def _settle(t):
    """gloo reads and writes device memory from the host without looking at any stream (rehearsals of the N > 1 path on
    one GPU, tools/rehearse_multi_rank.sh): drain the device before and after its transfers.  RCCL is stream-ordered."""
    import torch
    import torch.distributed as dist
    if getattr(t, "is_cuda", False) and dist.get_backend() == "gloo":
        torch.cuda.synchronize()


def partial_shape(dist: DistributionParams, viewports: int):
    w, h = get_distribution_target_size(dist)
    return (viewports, h, w, 4)


def gather_to_display(color, dists: List[DistributionParams], rank: int, world_size: int, viewports: int,
                      recv_buffers: Dict[int, object]):
    """Rank 0 returns {peer: tensor with the peer's partial frame}; other ranks return {}."""
    import torch
    import torch.distributed as dist
    if world_size == 1:
        return {}
    _settle(color)
    if rank == 0:
        ops = recv_buffers.get("ops")       # the receive list of the previous frame, while the shapes stay what they were
        fresh = ops is None
        for r in range(1, world_size):
            shape = partial_shape(dists[r], viewports)
            buf = recv_buffers.get(r)
            if buf is None or tuple(buf.shape) != shape:
                recv_buffers[r] = torch.empty(shape, dtype=torch.float32, device=color.device)
                fresh = True
        if fresh:
            ops = [dist.P2POp(dist.irecv, recv_buffers[r], r) for r in range(1, world_size)]
            recv_buffers["ops"] = ops
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        _settle(color)
        return {r: recv_buffers[r] for r in range(1, world_size)}
    for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, color, 0)]):
        q.wait()
    _settle(color)
    return {}


class LocalExchange:
    """The same exchange for several ranks that live in ONE process on one or more devices (the reference's own
    arrangement, and `--fake-devices`): a "send" is a device-to-device copy enqueued on the sending device's default stream
    into the display rank's receive buffer for that peer (trhip_copy_peer, which is what an RCCL send/recv pair amounts to over
    xGMI), followed by trhip_stream_wait_peer: the display device's default stream - where the stitch runs - waits for that copy.
    With every rank on one device (the fake-device tests) both are the same stream and the wait is a no-op.  A "receive" is
    nothing at all.  No host synchronisation anywhere: a frame is correct only if the renderer's stream dependencies around
    the exchange are (tests/test_gpu_parity.py::test_in_process_ranks_exchange_is_stream_ordered).  Call order per frame:
    every non-display rank's render(), then rank 0's."""

    def __init__(self, world_size: int):
        self.world_size = world_size
        self.mailbox = {}       # peer -> (DeviceBuffer on the display device, shape)
        self.display_ctx = None

    def attach(self, rank: int, ctx):
        if rank == 0:
            self.display_ctx = ctx

    def gather_to_display(self, color, dists: List[DistributionParams], rank: int, world_size: int, viewports: int, recv_buffers, ctx):
        from . import _lib
        if world_size == 1:
            return {}
        if rank == 0:
            return {r: self.mailbox[r][0] for r in range(1, world_size) if r in self.mailbox}
        if self.display_ctx is None:
            raise RuntimeError("LocalExchange: the display rank has to be created first")
        shape = partial_shape(dists[rank], viewports)
        nbytes = shape[0] * shape[1] * shape[2] * 16
        box = self.mailbox.get(rank)
        if box is None or box[1] != shape:
            self.display_ctx.sync()     # a stitch of the previous shape may still read the old buffer
            box = (self.display_ctx.alloc(max(nbytes, 16)), shape)
            self.mailbox[rank] = box
        if nbytes:
            rc = _lib.lib().trhip_copy_peer(self.display_ctx.h, box[0].data_ptr(), ctx.h, color.data_ptr(), nbytes, None)
            if rc:
                raise RuntimeError(_lib.lib().trhip_last_error().decode())
            # ranks on different devices: the copy sits on the source device's default stream, the stitch on the display device's
            rc = _lib.lib().trhip_stream_wait_peer(self.display_ctx.h, None, ctx.h, None)
            if rc:
                raise RuntimeError(_lib.lib().trhip_last_error().decode())
        return {}


class StandaloneExchange:
    """Calibration only (bench.py's load-balancer phase): nothing travels.  The display rank stitches standing buffers of the
    shapes it would receive, the other ranks send nothing, so every rank runs its share of a frame - the display rank with its
    stitch and tonemap - uncoupled from the others, and its wall time per frame says what that share costs it."""

    def __init__(self):
        self.mailbox = {}

    def attach(self, rank: int, ctx):
        pass

    def gather_to_display(self, color, dists: List[DistributionParams], rank: int, world_size: int, viewports: int, recv_buffers, ctx):
        if rank != 0:
            return {}
        out = {}
        for r in range(1, world_size):
            shape = partial_shape(dists[r], viewports)
            box = self.mailbox.get(r)
            if box is None or box[1] != shape:
                ctx.sync()          # a stitch of the previous shape may still read the old buffer
                box = (ctx.alloc(max(shape[0] * shape[1] * shape[2] * 16, 16)).zero(), shape)
                self.mailbox[r] = box
            out[r] = box[0]
        return out


def shard_viewports(viewports: int, rank: int, world_size: int) -> List[int]:
    """View sharding: viewport v belongs to device v mod N."""
    return list(range(rank, viewports, world_size))


def gather_views_to_display(local_views, viewports: int, rank: int, world_size: int, out=None):
    """Optional last step of a view-sharded frame: the finished views [k, H, W, 4] of every rank travel to rank 0, which
    returns the full [viewports, H, W, 4] stack (other ranks: None).  One grouped send/recv, no reduction: a view is
    owned by exactly one rank."""
    import torch
    import torch.distributed as dist
    if world_size == 1:
        return local_views
    _settle(local_views)
    if rank == 0:
        shape = (viewports,) + tuple(local_views.shape[1:])
        if out is None or tuple(out.shape) != shape:
            out = torch.empty(shape, dtype=local_views.dtype, device=local_views.device)
        mine = shard_viewports(viewports, 0, world_size)
        if mine:
            out[mine[0]::world_size] = local_views
        ops, staged = [], {}
        for r in range(1, world_size):
            k = len(shard_viewports(viewports, r, world_size))
            if k == 0:
                continue
            staged[r] = torch.empty((k,) + tuple(local_views.shape[1:]), dtype=local_views.dtype, device=local_views.device)
            ops.append(dist.P2POp(dist.irecv, staged[r], r))
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        _settle(local_views)
        for r, buf in staged.items():
            out[r::world_size] = buf
        return out
    if local_views.shape[0] > 0:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_views, 0)]):
            q.wait()
    return None


def reduce_samples_to_display(color, rank: int, world_size: int):
    """Sample sharding: every rank holds the mean of its own samples of every pixel, all ranks took the
    same number; one reduce(sum) to rank 0 (RCCL ring over xGMI), which divides by N.  In place on `color`."""
    import torch.distributed as dist
    if world_size == 1:
        return color
    _settle(color)
    dist.reduce(color, dst=0, op=dist.ReduceOp.SUM)
    _settle(color)
    if rank == 0:
        color.mul_(1.0 / world_size)
        return color
    return None
