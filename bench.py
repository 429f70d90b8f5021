#!/usr/bin/env python3
"""Benchmark of the path_tracer_stage hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload sponza_teapots|sponza_class|test_glb]

With N > 1 and no launcher in the environment (no RANK) the script starts its own ranks: it re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one process per GPU.
The default workload is sponza_teapots (BASELINE config 4's 1 M-triangle scene, the one north_star's target is stated on) for
every N; test_glb is BASELINE config 2.

A "step" is one frame: every pass of path_tracer_stage over one 1920x1080 image (1 spp, 4 bounces, all other
options at the reference's CLI defaults), followed by the multi-GPU gather/stitch and the tonemap - the
`render()` sequence of rt_renderer (reference src/rt_renderer.cc:84-133).  Scene upload, BVH build and image
save are outside the timed region (reference README "Benchmarking", docs/MANUAL.md:405-408).

Prints ONE JSON line (rank 0).  `value` = rays actually traced (closest-hit + shadow, device counters) per second
over all GPUs.  With N > 1 the pixels of the frame are sharded - shuffled strips (DISTRIBUTION_SHUFFLED_STRIPS, the
reference's command-line default) whose shares its load balancer settles during the untimed frames, or interleaved
scanlines with --strategy scanline - and the partial frames are gathered on rank 0 over RCCL: total work is fixed, so
scaling is "strong".

The K timed frames are K distinct frames (frame index = sample counter), four of them in flight at a time on their own
streams like the reference's frame slots (--frames-in-flight); the region is closed by a
synchronisation of every stream (+ barrier), so `ms_per_step` is elapsed / K and `frame_latency_ms` is what one frame
takes with a host sync after each.  Before the W warm-up steps --prewarm untimed frames bring a cold box to its clocks.
The `roofline` object is measured in a separate serialised re-run (one frame at a time, per-kernel HIP events), the
`cpu_baseline` by the CPU oracle on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Hardware queues the HIP runtime spreads streams over (default 4).  Six frame slots of a small shard want their own queue
# each; measured neutral for full frames.  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sponza_teapots", choices=["test_glb", "sponza_class", "sponza_teapots"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--sampler", type=int, default=0)
    ap.add_argument("--views", type=int, default=1, help="camera-grid viewports per frame (45 = the 5x9 light field of config 5)")
    ap.add_argument("--shard", default="pixels", choices=["pixels", "views", "samples"],
                    help="what N GPUs divide: scanlines of one frame (default, the reference's strategy), viewports, or samples")
    ap.add_argument("--strategy", default="auto", choices=["auto", "scanline", "strips"],
                    help="how N > 1 GPUs divide the pixels of a frame: interleaved scanlines, or shuffled strips whose shares the load "
                         "balancer sets (the reference's command-line default, src/tauray.cc:519-521); auto = strips")
    ap.add_argument("--no-balance", action="store_true", help="shuffled strips with equal shares: no load-balancer updates during the untimed frames")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="consecutive frames per path-tracing launch (trhip_pt_set_frame_batch); 0 = 2 on one GPU (1 if --steps is odd), "
                         "and for N > 1 pixel shards - whose launches are too small to fill a GPU - the largest of 5, 4, 3, 2 that divides --steps")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frame slots rendering concurrently (the reference keeps 2, src/context.hh:26); 1 = one frame at a time; "
                         "0 = 4 (on eight hardware queues; measured best from whole frames down to 1/8 shards, tools/shard_share_probe.py)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="all ranks on HIP device 0 (rehearsal on a one-GPU box, with --dist-backend gloo)")
    ap.add_argument("--save-display", default=None, help="rank 0 writes the last tonemapped frame to this .npy file")
    ap.add_argument("--prewarm", type=int, default=120, help="untimed frames before the warm-up steps (clocks, page faults)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, meet in one all-reduce, print the world size and stop (checks the launcher path without a GPU)")
    ap.add_argument("--sustained-frames", type=int, default=400,
                    help="N = 1: a second, longer timed region of this many frames reported as `sustained` (K = 20 frames are 0.1 s; "
                         "0 switches it off)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node and hand over their exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    # A multi-rank run that stops making progress (a peer died, an exchange deadlocked) dumps every thread's stack and exits
    # instead of hanging until somebody kills it; a whole run takes a minute or two.  TRHIP_BENCH_WATCHDOG overrides (seconds).
    watchdog = int(os.environ.get("TRHIP_BENCH_WATCHDOG", "900" if args.gpus > 1 else "0"))
    if watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)
    from tauray_amd import renderer as R
    from tauray_amd import scenes
    from tauray_amd.distribution import DISTRIBUTION_SCANLINE, DISTRIBUTION_SHUFFLED_STRIPS, LoadBalancer

    if args.frames_in_flight <= 0:
        args.frames_in_flight = 4
    world = args.gpus
    rank = 0
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        rank = int(os.environ.get("RANK", "0"))
        local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", str(rank)))
        world = int(os.environ.get("WORLD_SIZE", str(world)))
        if args.rendezvous_only:
            dist.init_process_group(args.dist_backend if args.dist_backend != "nccl" or torch.cuda.is_available() else "gloo")
            t = torch.tensor([rank + 1], dtype=torch.int64)
            dist.all_reduce(t)
            if rank == 0:
                print(json.dumps({"rendezvous": world, "rank_sum": int(t.item())}))
            dist.destroy_process_group()
            return
        if torch.cuda.device_count() > 0:
            local_rank %= torch.cuda.device_count()     # a launcher that narrows *_VISIBLE_DEVICES per rank leaves one device, index 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        local_rank = 0

    W, H = args.width, args.height
    scene = scenes.WORKLOADS[args.workload](W, H)
    if args.views > 1:      # light-field grid (src/tauray.cc:680-727): spacing 0.02, recentering distance 5
        from tauray_amd.scene import generate_camera_grid
        gw = 9 if args.views == 45 else args.views
        scene.cameras = generate_camera_grid(scene.cameras[0], gw, args.views // gw, 0.02, 0.02, 5.0)
    ctx = R.Context(local_rank)
    opt = R.options_for_scene(scene, max_bounces=args.bounces, samples_per_pixel=args.spp, samples_per_pass=1, sampler=args.sampler)
    strips = world > 1 and args.shard == "pixels" and args.strategy != "scanline"
    # A rank of a pixel-sharded job traces 1 / N of a frame per launch: at N = 8 a launch no longer fills the GPU, and several
    # frames per launch cost 12 % less per frame (tools/shard_share_probe.py).  The frames are the same
    # frames (tests/test_gpu_parity.py::test_frame_batches_render_the_frames_of_separate_calls).
    B = args.frames_per_launch
    if B <= 0:
        if world > 1 and args.shard == "pixels" and args.views == 1:
            B = next((b for b in (5, 4, 3, 2) if args.steps % b == 0), 1)
        else:       # whole frames: two per launch are worth 2 % on sponza_teapots and 5 % on test.glb, more are not
            B = 2 if (world == 1 and args.views == 1 and args.spp == 1 and args.steps % 2 == 0) else 1
    if args.steps % B:
        raise SystemExit(f"--steps {args.steps} is not a whole number of launches of {B} frames")
    rr = R.RtRenderer(ctx, scene, opt, (W, H), strategy=DISTRIBUTION_SHUFFLED_STRIPS if strips else DISTRIBUTION_SCANLINE, rank=rank, world_size=world,
                      viewports=args.views, shard=args.shard, frames_in_flight=args.frames_in_flight, frames_per_launch=B)

    def sync_all():
        rr.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def run_frames(n, one_at_a_time=False):
        for _ in range((n + B - 1) // B):      # B frames per render(); untimed regions round up
            rr.reset_accumulation()     # offline frames: accumulation reset, sample counter kept (src/tauray.cc:1101)
            rr.render()
            if one_at_a_time:           # per-kernel timing wants kernels that own the chip: no second frame next to them
                rr.sync()

    # ---- timed region: W warm-up frames, then exactly K frames between barriers
    rr.set_profiling(False, False)
    # Not part of the W warm-up steps: a fresh box takes a few hundred milliseconds of work to reach its clocks and to fault
    # in every buffer, more than W = 3 frames of 2 ms give it.  A fixed frame count keeps the ranks of a multi-GPU job in step.
    balance = None
    if strips and not args.no_balance and args.prewarm >= 16:
        # load_balancer (src/load_balancer.cc:12-32, updated once per frame by src/tauray.cc:1005-1116) during the untimed frames:
        # share / time = a device's speed, the shares move towards speed / sum of speeds by the reference's 0.1 EMA step.  The
        # reference times "path tracing" on every device; with frames in flight that timer spans several overlapping frames and
        # says little about what a share costs (tools/display_rank_probe.py), so a rank's time is the wall time per frame of its
        # own work running free - path tracing of its share and, on the display rank, the stitch and the tonemap next to it
        # (transfer.StandaloneExchange: nothing travels during these frames).  The display rank ends up with less than 1 / N.
        # Reading the clocks drains the frame slots, which the timed frames should not pay for: the shares then stay.
        from tauray_amd.transfer import StandaloneExchange
        lb = LoadBalancer(world)
        every, rounds = 8, 24
        rr.exchange = StandaloneExchange()
        run_frames(max(args.prewarm - every * rounds, 8))
        for _ in range(rounds):
            rr.sync()
            t1 = time.perf_counter()
            run_frames(every)
            rr.sync()
            times = [0.0] * world
            dist.all_gather_object(times, (time.perf_counter() - t1) / (((every + B - 1) // B) * B) * 1e3)
            rr.set_device_workloads(lb.update(times))
        rr.exchange = None
        balance = {"updates": rounds, "frames_per_update": every, "workloads": [round(w, 4) for w in lb.workloads],
                   "ms_per_frame_running_free": [round(t, 4) for t in times]}
        sync_all()
        run_frames(args.frames_in_flight * 2)
    else:
        run_frames(args.prewarm)
    sync_all()
    rr.reset_accumulation(reset_sample_counter=True)
    run_frames(args.warmup)
    sync_all()
    rr.reset_counters()
    t0 = time.perf_counter()
    run_frames(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    counters = rr.counters()
    rays_local = counters["closest_rays"] + counters["shadow_rays"]

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_local], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = int(r.item())
    else:
        rays_total = rays_local

    if counters["stack_overflows"]:
        raise RuntimeError("BVH traversal stack overflow: results invalid")

    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_total / elapsed / 1e6
    result = {
        "metric": "Mray/s (closest-hit + shadow rays traced) @%dx%d, %d bounces, %d spp" % (W, H, args.bounces, args.spp),
        "value": round(mrays, 2), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if args.workload != "test_glb" else "reference fixture test/test.glb (81 364 triangles)",
        "config": {"workload": args.workload, "triangles": scene.triangle_count, "width": W, "height": H, "bounces": args.bounces,
                   "spp": args.spp, "sampler": ["uniform-random", "sobol-owen", "sobol-z2", "sobol-z3"][args.sampler],
                   "parallelism": ({"pixels": ("shuffled strips x%d, balanced shares + RCCL gather" if balance else "shuffled strips x%d + RCCL gather") if strips
                                    else "scanline-sharded x%d + RCCL gather", "views": "view-sharded x%d, no exchange",
                                    "samples": "sample-sharded x%d + RCCL reduce"}[args.shard] % world) if world > 1 else "single GPU",
                   "views": args.views, "frames_in_flight": args.frames_in_flight, "frames_per_launch": B, "prewarm_frames": args.prewarm,
                   "scene_hash": scenes.scene_hash(scene)},
        "accel_build_ms": round(rr.scene_update.accel["build_ms"], 2),
        **({"load_balance": balance} if balance else {}),
        "rays_per_frame": rays_total // args.steps,
        "msample_per_s": round(W * H * args.views * args.spp * args.steps / elapsed / 1e6, 2),
    }

    # ---- frame latency distribution (mean and p50): the same frames again with a host sync after each one,
    # so unlike `ms_per_step` (back-to-back frames, the metric) this includes the enqueue latency of every frame
    if world == 1:
        # A caller that waits for every frame has no use for frame slots: the renderer of this loop has none, so each frame runs
        # as four concurrent lanes (the stage's automatic schedule) instead of one lane per slot.
        lone = R.RtRenderer(ctx, scene, opt, (W, H), strategy=DISTRIBUTION_SCANLINE, viewports=args.views, frames_in_flight=1) if args.frames_in_flight > 1 else rr
        lone.set_profiling(False, False)
        for _ in range(10):
            lone.reset_accumulation(); lone.render()
        lone.sync()
        lat = []
        for _ in range(min(args.steps, 50)):
            t1 = time.perf_counter()
            lone.reset_accumulation()
            lone.render()
            lone.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
        if lone is not rr:
            lone.close()
        lat.sort()
        result["frame_latency_ms"] = {"p50": round(lat[len(lat) // 2], 4), "mean": round(sum(lat) / len(lat), 4), "min": round(lat[0], 4),
                                      "frames": len(lat), "note": "host sync after every frame; renderer without frame slots (four lanes per frame)"}
        # SURVEY.md 8(d) / BASELINE.md define ms/frame as host wall time around one render() including the stream sync: the same
        # metric by that definition (one frame at a time, no frames in flight), beside the pipelined `value`
        p50 = lat[len(lat) // 2]
        result["ms_per_frame_sync"] = round(p50, 4)
        result["value_sync_per_frame"] = round(result["rays_per_frame"] / (p50 * 1e-3) / 1e6, 2)
        if args.sustained_frames > 0:      # a timed region long enough for a 1 Hz utilisation sampler to see
            sync_all()
            t1 = time.perf_counter()
            run_frames(args.sustained_frames)
            sync_all()
            dt = time.perf_counter() - t1
            result["sustained"] = {"frames": args.sustained_frames, "ms_per_frame": round(dt / args.sustained_frames * 1e3, 4),
                                   "value": round(result["rays_per_frame"] * args.sustained_frames / dt / 1e6, 2), "unit": "Mray/s"}

    # ---- roofline of the dominant kernel (k_trace_closest), rank 0
    if not args.no_roofline:
        # every rank takes part (a frame of a multi-GPU job ends in an exchange between the ranks); rank 0's kernels are reported.
        # re-run of the identical frames (same frame indices) with per-kernel HIP events; detailed timing serialises the
        # frame (no shadow/closest overlap), so every kernel is measured owning the chip (instance k_trace_closest<false, true>)
        rr.set_profiling(False, True)
        rr.reset_accumulation(reset_sample_counter=True)
        run_frames(args.warmup, True)
        rr.reset_counters()
        run_frames(args.steps, True)
        timings = rr.timings()
        launches = max(timings["trace_closest_launches"], 1)
        avg_ms = timings["trace_closest_ms"] / launches
        # counted re-run for the algorithmic byte model
        rr.set_profiling(True, False)
        rr.reset_accumulation(reset_sample_counter=True)
        run_frames(args.warmup)
        rr.reset_counters()
        run_frames(args.steps)
        c = rr.counters()
        # bytes per SURVEY.md section 8(d), restricted to what trace kernels touch; the closest-hit kernel's share of
        # node/triangle work is apportioned by ray count (both trace kernels walk the same structure)
        closest_share = c["closest_rays"] / max(c["closest_rays"] + c["shadow_rays"], 1)
        node_bytes = rr.scene_update.accel["node_bytes"]   # what one node visit reads (trhip_accel_info)
        trace_bytes = (c["node_visits"] * node_bytes + c["tri_tests"] * 48 + c["alpha_tests"] * 52) * closest_share \
            + c["closest_rays"] * (16 + 16 + 16 + 16)   # ray origin + direction + misc read, hit record write
        bytes_per_launch = trace_bytes / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        frame_bytes = (c["node_visits"] * node_bytes + c["tri_tests"] * 48 + c["alpha_tests"] * 52 + c["surface_hits"] * 268
                       + (c["closest_rays"] + c["shadow_rays"]) * (2 * 48 + 2 * 20) + W * H * args.views * args.spp * 16 * args.steps) / args.steps
        # HBM traffic of the same kernel from the committed PMC passes (tools/profile_round.sh; FETCH_SIZE + WRITE_SIZE in
        # separate runs).  Lower bound as reported; FETCH_SIZE may under-report by up to 2x on gfx950 (upper bound given too).
        traffic, traffic_range, traffic_src, valu = None, None, None, None
        try:
            import glob, json as _json
            for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*", "pmc_summary.json")), reverse=True):
                k = _json.load(open(f)).get(args.workload, {}).get("k_trace_closest", {})
                if "hbm_traffic_bytes_per_launch_range" in k:
                    traffic_range = k["hbm_traffic_bytes_per_launch_range"]
                    traffic, traffic_src = traffic_range[0], os.path.relpath(f, os.path.dirname(os.path.abspath(__file__)))
                    # HBM is not what binds this kernel (the tree is served by L2 / Infinity Cache): the VALU counters of the
                    # same committed passes say what does
                    valu = {"issue_busy": k.get("valu_issue_busy"), "lane_utilisation": k.get("valu_lane_utilisation"),
                            "wait_fraction": k.get("wait_fraction"), "source": traffic_src}
                    break
        except Exception:
            pass
        result["roofline"] = {
            "bound": "hbm", "kernel": "k_trace_closest", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_range": traffic_range, "traffic_source": traffic_src,
            "valu": valu,
            # what the PMC traffic says the kernel really pulls from HBM, as a fraction of peak (the tree is served by L2 / MALL, so
            # this is far below `frac`, which prices the algorithmic bytes): HBM is not what binds this kernel
            "traffic_frac": [round(t / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for t in traffic_range] if (traffic_range and avg_ms > 0) else None,
            "avg_launch_ms": round(avg_ms, 4), "launches": launches, "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "frame_algorithmic_GBps": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            # what a frame cannot avoid moving even with a perfectly cached tree: ray + hit records written and read once, one
            # framebuffer write
            "frame_compulsory_GBps": round(((c["closest_rays"] + c["shadow_rays"]) * (2 * 48 + 2 * 20) + W * H * args.views * args.spp * 16 * args.steps)
                                           / args.steps / (ms_per_step * 1e-3) / 1e9, 1),
            "kernel_ms_per_frame": {k: round(timings[k + "_ms"] / args.steps, 4) for k in ("trace_closest", "trace_shadow", "shade", "raygen", "resolve")},
            "node_visits_per_ray": round(c["node_visits"] / max(c["closest_rays"] + c["shadow_rays"], 1), 2),
            "tri_tests_per_ray": round(c["tri_tests"] / max(c["closest_rays"] + c["shadow_rays"], 1), 2),
        }

    # ---- CPU baseline: the oracle (a port; the reference has no CPU path) on the host cores, rank 0, N = 1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as B
        osc = B.OracleScene(scene)
        oopt = B.options_for_scene(scene, max_bounces=args.bounces, samples_per_pixel=args.spp, sampler=args.sampler)
        cores = os.cpu_count() or 1
        frames = 0
        t0 = time.perf_counter()
        while True:
            osc.render_pt(oopt, W, H, frame_counter=frames, threads=cores, viewports=args.views)
            frames += 1
            if time.perf_counter() - t0 >= args.cpu_seconds:
                break
        dt = time.perf_counter() - t0
        oc = osc.counters()
        result["cpu_baseline"] = {
            "value": round((oc["closest_rays"] + oc["shadow_rays"]) / dt / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"{frames} full {W}x{H} frame(s) of the same workload, {dt:.1f} s of wall time, OpenMP over rows",
            "ms_per_frame": round(dt / frames * 1e3, 1),
        }

    if args.save_display:       # frame 0 again on every rank, outside all timing
        rr.set_profiling(False, False)
        rr.reset_accumulation(reset_sample_counter=True)
        rr.render()
        sync_all()
        if rank == 0:
            np.save(args.save_display, rr.download("display")[:args.views])        # frame 0 of the launch
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
