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
    if rank == 0:
        ops = []
        for r in range(1, world_size):
            shape = partial_shape(dists[r], viewports)
            buf = recv_buffers.get(r)
            if buf is None or tuple(buf.shape) != shape:
                buf = torch.empty(shape, dtype=torch.float32, device=color.device)
                recv_buffers[r] = buf
            ops.append(dist.P2POp(dist.irecv, buf, r))
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        return {r: recv_buffers[r] for r in range(1, world_size)}
    for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, color, 0)]):
        q.wait()
    return {}
