"""N > 1 path on CPU: world_size-2 and -3 `gloo` jobs exercise the distribution math and the transfer protocol
(tauray_amd/transfer.py) with the CPU oracle standing in for the per-rank renderer and a numpy restatement of
the stitch shaders standing in for the stitch kernel.  The stitched frame must equal the single-device frame
bit for bit (pixel -> RNG mapping is by absolute pixel coordinate, shader/rt.glsl:181-196)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT

W, H = 96, 64


def _stitch_numpy(primary, partial, d, b):
    """shader/stitch_scanline.comp:20-50 / stitch_shuffled_strips.comp:20-63 for one partial image."""
    from tauray_amd import distribution as D
    if d.strategy == D.DISTRIBUTION_SCANLINE:
        rows = partial.shape[1]
        primary[:, d.index::d.count][:, :rows] = partial
    else:
        for p in range(d.count):
            j = D.permute_region_id(d.index + p, d.size, b)
            if j < d.size[0] * d.size[1]:
                primary[:, j // d.size[0], j % d.size[0]] = partial[:, p // d.size[0], p % d.size[0]]


def _worker(rank, world, strategy, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as B
    from tauray_amd import distribution as D
    from tauray_amd.gltf import load_glb
    from tauray_amd.transfer import gather_to_display, partial_shape
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    osc = B.OracleScene(scene)
    opt = B.options_for_scene(scene, max_bounces=3)
    ratios = [1.0 / world] * world
    dists, cum = [], 0.0
    for i in range(world):
        dists.append(D.get_device_distribution_params((W, H), strategy, cum, ratios[i], i, world, i == 0))
        cum += ratios[i]
    d = dists[rank]
    tw, th = D.get_distribution_target_size(d)
    dc = B.DistributionC(W, H, strategy, d.index, d.count, 1 if d.primary else 0)
    color = osc.render_pt(opt, W, H, dist=dc, target_size=(tw, th), threads=2)
    t = torch.from_numpy(color)
    assert tuple(t.shape) == partial_shape(d, 1)
    parts = gather_to_display(t, dists, rank, world, 1, {})
    if rank == 0:
        b = D.calculate_shuffled_strips_b((W, H))
        full = color.copy()
        for r, buf in parts.items():
            _stitch_numpy(full, buf.numpy(), dists[r], b)
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,strategy", [(2, 1), (3, 1), (2, 2)])
def test_sharded_frame_equals_single_device_frame(tmp_path, world, strategy, oracle):
    from tauray_amd.gltf import load_glb
    out = str(tmp_path / "full.npy")
    port = 29500 + (os.getpid() + world * 7 + strategy) % 2000
    mp.spawn(_worker, args=(world, strategy, port, out), nprocs=world, join=True)
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    ref = oracle.OracleScene(scene).render_pt(oracle.options_for_scene(scene, max_bounces=3), W, H)
    got = np.load(out)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), f"{(got != ref).any(-1).sum()} pixels differ"


def _shard_worker(rank, world, mode, port, out_path):
    """View / sample sharding with the oracle as the per-rank renderer."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as B
    from tauray_amd import scene as S
    from tauray_amd.gltf import load_glb
    from tauray_amd.transfer import gather_views_to_display, reduce_samples_to_display, shard_viewports
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    scene.cameras = S.generate_camera_grid(scene.cameras[0], VIEWS, 1, 0.3, 0.3, 5.0)
    osc = B.OracleScene(scene)
    if mode == "views":
        mine = shard_viewports(VIEWS, rank, world)
        osc.set_shard(viewport_base=rank, viewport_stride=world)
        opt = B.options_for_scene(scene, max_bounces=2)
        local = osc.render_pt(opt, W, H, viewports=len(mine), threads=2) if mine else np.zeros((0, H, W, 4), np.float32)
        full = gather_views_to_display(torch.from_numpy(local), VIEWS, rank, world)
        if rank == 0:
            np.save(out_path, full.numpy())
    else:
        osc.set_shard(sample_base=rank, sample_stride=world)
        opt = B.options_for_scene(scene, max_bounces=2, samples_per_pixel=SPP // world)
        local = osc.render_pt(opt, W, H, frame_counter=1, threads=2)
        total = reduce_samples_to_display(torch.from_numpy(local), rank, world)
        if rank == 0:
            np.save(out_path, total.numpy())
    dist.barrier()
    dist.destroy_process_group()


VIEWS, SPP = 5, 6


@pytest.mark.parametrize("world,mode", [(2, "views"), (3, "views"), (6, "views"), (2, "samples"), (3, "samples")])
def test_view_and_sample_shards_equal_single_device(tmp_path, world, mode, oracle):
    from tauray_amd import scene as S
    from tauray_amd.gltf import load_glb
    out = str(tmp_path / "full.npy")
    port = 31500 + (os.getpid() + world * 11 + len(mode)) % 2000
    mp.spawn(_shard_worker, args=(world, mode, port, out), nprocs=world, join=True)
    scene = load_glb(os.path.join(GOLDEN, "test.glb"), W, H)
    scene.cameras = S.generate_camera_grid(scene.cameras[0], VIEWS, 1, 0.3, 0.3, 5.0)
    osc = oracle.OracleScene(scene)
    got = np.load(out)
    if mode == "views":     # more ranks than views (6 > 5): the last rank owns nothing
        ref = osc.render_pt(oracle.options_for_scene(scene, max_bounces=2), W, H, viewports=VIEWS)
        assert got.shape == ref.shape and np.array_equal(got, ref)
    else:                   # same samples, summed in a different order
        ref = osc.render_pt(oracle.options_for_scene(scene, max_bounces=2, samples_per_pixel=SPP), W, H, frame_counter=1)
        assert float(np.abs(got[..., :3] - ref[..., :3]).max()) <= 2e-6 * max(1.0, float(np.abs(ref[..., :3]).max()))


def _balance_worker(rank, world, port, out_path):
    """bench.py's load-balancer rounds with a made-up cost model in place of the GPU: a rank's frame costs a fixed part (larger
    on the display rank, which also stitches and tonemaps) plus a part proportional to its share of the pixels."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tauray_amd import distribution as D
    lb = D.LoadBalancer(world)
    fixed = 0.21 if rank == 0 else 0.12
    history = []
    for _ in range(60):
        mine = fixed + 3.86 * lb.workloads[rank]
        times = [0.0] * world
        dist.all_gather_object(times, mine)
        lb.update(times)
        history.append(list(lb.workloads))
    # what every rank would hand to set_device_workloads: the strips of the whole image, each exactly once
    size = (1920, 1080)
    cum, strips = 0.0, []
    for i in range(world):
        d = D.get_device_distribution_params(size, D.DISTRIBUTION_SHUFFLED_STRIPS, cum, lb.workloads[i], i, world, i == 0)
        cum += lb.workloads[i]
        strips.append((d.index, d.count))
    everyone = [None] * world
    dist.all_gather_object(everyone, (history[-1], strips, times))
    if rank == 0:
        np.save(out_path, np.array([len({repr(e[:2]) for e in everyone})] + list(history[-1]) + list(everyone[0][2]) + [c for s in strips for c in s], dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_load_balancer_rounds_agree_on_every_rank_and_even_out_the_frame_times(tmp_path):
    world = 4
    out = str(tmp_path / "balance.npy")
    port = 29500 + (os.getpid() + 977) % 2000
    mp.spawn(_balance_worker, args=(world, port, out), nprocs=world, join=True)
    r = np.load(out)
    assert r[0] == 1, "the ranks computed different shares from the same gathered times"
    shares, times, strips = r[1:1 + world], r[1 + world:1 + 2 * world], r[1 + 2 * world:].reshape(world, 2)
    assert abs(shares.sum() - 1) < 1e-9 and shares[0] < 1 / world < shares[1]
    assert times.max() / times.min() < 1.01, times          # equal frame times: the display rank's extra work is paid for by a smaller share
    # consecutive pixel ranges of whole strips that cover the image (src/distribution_strategy.cc:62-126)
    assert strips[0, 0] == 0 and all(strips[i, 0] + strips[i, 1] == strips[i + 1, 0] for i in range(world - 1))
    assert strips[-1, 0] + strips[-1, 1] >= 1920 * 1080
