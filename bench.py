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

Prints ONE JSON line (rank 0).

`value` follows SURVEY.md section 8(d) to the letter: Mray/s = rays actually traced (closest-hit + shadow, device counters) over
all GPUs / host wall time, where every frame is timed from before render() to after the stream sync - ONE FRAME AT A TIME, no
frames in flight, one frame per launch.  The timed region holds max(K, 50) such frames (`steps_effective`: K = 20 frames would
be 0.1 s) between barriers; `ms_per_step` = elapsed / steps_effective, `frame_ms` holds mean and p50 of the per-frame times.
`value_pipelined` is the same workload the way a renderer that does not wait runs it: four frames in flight on their own
streams like the reference's frame slots (src/context.hh:26), two frames per launch on one GPU (`pipelined` holds the details).
With N > 1 the pixels of the frame are sharded - shuffled strips (DISTRIBUTION_SHUFFLED_STRIPS, the reference's command-line
default) whose shares its load balancer settles during the untimed frames, or interleaved scanlines with --strategy scanline -
and the partial frames are gathered on rank 0 over RCCL: total work is fixed, so scaling is "strong".

`roofline` (rank 0): the dominant kernel, k_trace_closest, timed alone with HIP events on its launch stream, against every level
that could bound it - vector-instruction issue (peak = a v_fma_f32 loop measured in this run), L1 -> L2 request bytes, L2-miss
(fabric: Infinity Cache + HBM) bytes - from rocprofv3 counter passes this script runs itself (children of this process, each
pass its own run with the kernel trace only; byte-per-request factors from profiles/r3/calibration.json).  `bound` is the level
with the highest fraction, `frac` that fraction.  `algorithmic_GBps` is SURVEY.md 8(d)'s counted-work figure (cache served).
`cpu_baseline`: the CPU oracle on the host cores.  `parity`: frame 0 of the workload as the timed kernels render it against the oracle's
frame 0 at the same size (share of pixels outside 1 %, error of the mean, RMS): a kernel that skipped work fails the bench's own run.
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
# dmabuf IPC is what the host driver of these boxes supports: RCCL and hipIpcGetMemHandle across processes fail without it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); 6.3-6.9 TB/s measured with a float4 stream
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth (MI355X_MICROARCH.md section L2)
MIN_TIMED_FRAMES = 50   # mean and p50 over >= 50 frames with distinct frame indices

# rocprofv3 counter passes of the roofline (each its own run; SQ has 8 slots, TCC 4 - MI355X_MICROARCH.md section PMC slots)
PMC_PASSES = {
    "sq": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY",
           "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
    "tcc": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
    "l2": ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
    "l1": ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "GRBM_GUI_ACTIVE"],
}
HOT_KERNELS = {"k_trace_closest": "trace_closest", "k_trace_shadow": "trace_shadow", "k_shade": "shade"}


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
    ap.add_argument("--sampler", type=int, default=0, help="0 uniform-random, 1 sobol-owen, 2 sobol-z2, 3 sobol-z3")
    ap.add_argument("--preset", default=None, choices=["quality", "reference", "accumulation"],
                    help="the path-tracer options of one of the reference's presets (data/presets/*.cfg: film, ray depth, regularisation, "
                         "triangle-light mode), one sample per pixel and frame unless --spp says otherwise")
    ap.add_argument("--general-kernels", action="store_true", help="no shading program compiled for the option set (trhip_pt_set_specialization 0): A/B")
    ap.add_argument("--ieee-shading", action="store_true", help="shading kernels at IEEE fp32 (trhip_pt_set_shading_arithmetic 1): A/B")
    ap.add_argument("--views", type=int, default=1, help="camera-grid viewports per frame (45 = the 5x9 light field of config 5)")
    ap.add_argument("--shard", default="pixels", choices=["pixels", "views", "samples"],
                    help="what N GPUs divide: scanlines of one frame (default, the reference's strategy), viewports, or samples")
    ap.add_argument("--strategy", default="auto", choices=["auto", "scanline", "strips"],
                    help="how N > 1 GPUs divide the pixels of a frame: interleaved scanlines, or shuffled strips whose shares the load "
                         "balancer sets (the reference's command-line default, src/tauray.cc:519-521); auto = strips")
    ap.add_argument("--no-balance", action="store_true", help="shuffled strips with equal shares: no load-balancer updates during the untimed frames")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="pipelined region: consecutive frames per path-tracing launch (trhip_pt_set_frame_batch); 0 = 2 on one GPU, and for "
                         "N > 1 pixel shards - whose launches are too small to fill a GPU - 5")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="pipelined region: frame slots rendering concurrently (the reference keeps 2, src/context.hh:26); "
                         "0 = 4 (on eight hardware queues; measured best from whole frames down to 1/8 shards, tools/shard_share_probe.py)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "native", "torch", "ipc"],
                    help="what carries the partial frames of N > 1 pixel shards to rank 0: native = libtrhip_comm.so (include/trhip_comm.h: grouped "
                         "ncclSend / ncclRecv on RCCL), torch = torch.distributed point-to-point; auto = native with the nccl backend, torch otherwise")
    ap.add_argument("--one-device", action="store_true", help="all ranks on HIP device 0 (rehearsal on a one-GPU box, with --dist-backend gloo)")
    ap.add_argument("--save-display", default=None, help="rank 0 writes the last tonemapped frame to this .npy file")
    ap.add_argument("--prewarm", type=int, default=120, help="untimed frames before the warm-up steps (clocks, page faults)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="roofline without the rocprofv3 counter passes (issue rate and traffic stay null)")
    ap.add_argument("--pmc-timeout", type=float, default=75.0, help="seconds one counter pass may take")
    ap.add_argument("--pmc-child", action="store_true", help="internal: render a few serialised frames and exit (what the counter passes profile)")
    ap.add_argument("--pmc-dump", default=None, help="directory that keeps the per-kernel counter summary of the passes (profiles/)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, meet in one all-reduce, print the world size and stop (checks the launcher path without a GPU)")
    ap.add_argument("--sustained-frames", type=int, default=400,
                    help="N = 1: a longer pipelined region of this many frames reported as `sustained` (0 switches it off)")
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


def stage_options(args, scene, R):
    """The path tracer's options of this run: the command-line set of the reference with --bounces / --spp / --sampler, or one of
    the reference's presets (tauray_amd/presets.py), which brings its own ray depth."""
    kw = dict(max_bounces=args.bounces, samples_per_pixel=args.spp, samples_per_pass=1, sampler=args.sampler)
    if args.preset:
        from tauray_amd.presets import REFERENCE_PRESETS
        kw.update(REFERENCE_PRESETS[args.preset])
        kw["samples_per_pixel"] = args.spp
        args.bounces = kw["max_bounces"]
    if args.general_kernels:
        os.environ["TRHIP_SPECIALIZE"] = "0"
    if args.ieee_shading:
        os.environ["TRHIP_SHADE_FAST"] = "0"
    args.option_kw = kw
    return R.options_for_scene(scene, **kw)


def pmc_child(args):
    """What a counter pass profiles: the launches the roofline times - one lane, unfused, every kernel owning the chip
    (trhip_pt_set_profiling detailed timing: k_trace_closest<false, true, ..>), two frames per launch like the timed-alone run."""
    from tauray_amd import renderer as R
    from tauray_amd import scenes
    from tauray_amd.distribution import DISTRIBUTION_SCANLINE
    W, H = args.width, args.height
    scene = scenes.WORKLOADS[args.workload](W, H)
    if args.views > 1:      # the same camera grid as the timed run: the launch the counters describe is the launch the roofline times
        from tauray_amd.scene import generate_camera_grid
        gw = 9 if args.views == 45 else args.views
        scene.cameras = generate_camera_grid(scene.cameras[0], gw, args.views // gw, 0.02, 0.02, 5.0)
    ctx = R.Context(0)
    opt = stage_options(args, scene, R)
    B = max(args.frames_per_launch, 1)
    rr = R.RtRenderer(ctx, scene, opt, (W, H), strategy=DISTRIBUTION_SCANLINE, frames_in_flight=1, frames_per_launch=B, viewports=args.views)
    rr.set_profiling(False, True)
    for _ in range(max(args.steps // B, 1) + 1):     # the first launch is a warm-up like any other: whole frames either way
        rr.reset_accumulation()
        rr.render()
        rr.sync()
    rr.close()


def run_pmc_passes(args, B, dump_dir=None):
    """Runs this script as --pmc-child under `rocprofv3 --pmc <set> --kernel-trace`, one run per counter set, and returns
    ({kernel: {counter: average per launch, 'launches': n, 'avg_us_<pass>': duration under that pass}}, error or None)."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {}, "rocprofv3 not found"
    out = collections.defaultdict(dict)
    errors = []
    tmp = tempfile.mkdtemp(prefix="trhip_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--width", str(args.width), "--height", str(args.height),
             "--bounces", str(args.bounces), "--spp", str(args.spp), "--sampler", str(args.sampler), "--steps", "4", "--frames-per-launch", str(B), "--views", str(args.views)] + \
            (["--preset", args.preset] if args.preset else []) + (["--general-kernels"] if args.general_kernels else []) + (["--ieee-shading"] if args.ieee_shading else [])
    for name, counters in PMC_PASSES.items():
        d = os.path.join(tmp, name)
        cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--"] + child
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=args.pmc_timeout)
        except subprocess.TimeoutExpired:
            errors.append(f"{name}: timed out after {args.pmc_timeout:.0f} s")
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            errors.append(f"{name}: rc {p.returncode}: " + p.stdout.decode("utf-8", "replace")[-300:].replace("\n", " | "))
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = collections.defaultdict(set)
        dur = collections.defaultdict(float)
        for r in csv.DictReader(open(files[0])):
            kn = r["Kernel_Name"]
            m = re.search(r"(k_[a-z_0-9]+)<", kn)
            key = m.group(1) if m and m.group(1) in HOT_KERNELS else None
            if key is None and kn.startswith("trhip_spec_shade"):      # the shading program compiled for the option set (csrc/shade_spec.hip)
                key = "k_shade"
            if key is None:
                continue
            if key == "k_trace_closest" and "<false, true" not in kn:      # the timed-alone instance only
                continue
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen[key]:
                seen[key].add(r["Dispatch_Id"])
                dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k in agg:
            n = len(seen[k])
            out[k]["launches"] = n
            out[k]["avg_us_" + name] = round(dur[k] / n / 1e3, 1)
            for c, v in agg[k].items():
                out[k][c] = v / n
    if dump_dir:
        os.makedirs(dump_dir, exist_ok=True)
        json.dump({k: {c: (round(v, 1) if isinstance(v, float) else v) for c, v in d.items()} for k, d in out.items()},
                  open(os.path.join(dump_dir, f"{args.workload}_pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)
    out["_frames"] = {"frames": (max(4 // B, 1) + 1) * B}      # frames the child rendered (pmc_child: steps 4, warm-up launch included)
    return dict(out), ("; ".join(errors) if errors else None)


def whole_frame_utilisation(pmc, frame_ms, valu_peak_ginst, l1_peak_gacc):
    """How full the chip is over a timed frame, from the counters of every launch of a frame: vector instructions, L1 line accesses
    and L2-miss bytes of the three hot kernels summed over the frame's launches, divided by the frame's wall time and the unit's peak.
    The counters come from the serialised launches of the counter passes (rocprofv3 collects per dispatch and therefore one dispatch
    at a time); instruction, access and request COUNTS do not depend on what else is running, so the sums hold for the overlapped launches
    of the timed frames too (whose closest-hit and shadow rays share a fused launch: same loops).  Busy cycles do depend on it and are
    given for the serialised launches only, as the share of the frame they would fill one after the other."""
    frames = (pmc.get("_frames") or {}).get("frames")
    if not frames or frame_ms <= 0:
        return None
    s = frame_ms * 1e-3
    tot = {"valu": 0.0, "l1": 0.0, "fabric": 0.0, "waves": 0.0, "busy_us": 0.0}
    have = set()
    for k in HOT_KERNELS:
        d = pmc.get(k)
        if not d:
            continue
        n = d.get("launches", 0) / frames      # launches of this kernel per frame
        if "SQ_INSTS_VALU" in d:
            tot["valu"] += d["SQ_INSTS_VALU"] * n; have.add("valu")
            tot["waves"] += d.get("SQ_WAVES", 0.0) * n
            tot["busy_us"] += d.get("avg_us_sq", 0.0) * n
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in d:
            tot["l1"] += d["TCP_TOTAL_CACHE_ACCESSES_sum"] * n; have.add("l1")
        if "TCC_EA0_RDREQ_sum" in d:
            rd32 = d.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            wr64 = d.get("TCC_EA0_WRREQ_64B_sum", 0.0)
            tot["fabric"] += ((d["TCC_EA0_RDREQ_sum"] - rd32) * 128.0 + rd32 * 32.0 + wr64 * 64.0 + (d.get("TCC_EA0_WRREQ_sum", 0.0) - wr64) * 32.0) * n; have.add("fabric")
    out = {"frame_ms": round(frame_ms, 4)}
    if "valu" in have and valu_peak_ginst:
        out["valu_frac"] = round(tot["valu"] / s / 1e9 / valu_peak_ginst, 4)
        out["vector_instructions_per_frame"] = int(tot["valu"])
        out["waves_per_frame"] = int(tot["waves"])
        out["serialised_kernel_time_over_frame_time"] = round(tot["busy_us"] * 1e-3 / frame_ms, 3)
    if "l1" in have and l1_peak_gacc:
        out["l1_frac"] = round(tot["l1"] / s / 1e9 / l1_peak_gacc, 4)
    if "fabric" in have:
        out["fabric_frac"] = round(tot["fabric"] / s / 1e9 / HBM_PEAK_GBS, 4)
        out["fabric_bytes_per_frame"] = int(tot["fabric"])
    return out


def level_fractions(pmc_k, avg_ms, valu_peak_ginst, l1_peak_gacc=None):
    """Per-level rates of one kernel from its per-launch counters and its un-profiled launch time.  Factors: profiles/r3/calibration.json
    (one TCP_TCC_READ_REQ = one 128-byte line; one TCC_EA0_RDREQ = 128 bytes unless counted as _32B; write requests 64 / 32 bytes)."""
    s = avg_ms * 1e-3
    lv = {}
    if not pmc_k or s <= 0:
        return lv
    if "SQ_INSTS_VALU" in pmc_k and valu_peak_ginst:
        g = pmc_k["SQ_INSTS_VALU"] / s / 1e9
        lv["valu"] = {"achieved": round(g, 1), "peak": round(valu_peak_ginst, 1), "unit": "Ginst/s", "frac": round(g / valu_peak_ginst, 4),
                      "insts_per_launch": int(pmc_k["SQ_INSTS_VALU"]),
                      "wait_fraction": round(pmc_k["SQ_WAIT_ANY"] / pmc_k["SQ_WAVE_CYCLES"], 3) if pmc_k.get("SQ_WAVE_CYCLES") else None,
                      "hw_lanes_per_inst": round(pmc_k["SQ_THREAD_CYCLES_VALU"] / pmc_k["SQ_INSTS_VALU"], 1) if pmc_k.get("SQ_THREAD_CYCLES_VALU") else None}
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc_k and l1_peak_gacc:
        # the vector L1 (TCP) of a CU takes one cache-line (tag) access per clock (tools/ubench/l1_tags.hip; the peak is measured in
        # this run by trhip_calibrate_l1); a lane's seven 16-byte loads of one node are seven accesses to one line
        a = pmc_k["TCP_TOTAL_CACHE_ACCESSES_sum"] / s / 1e9
        lv["l1"] = {"achieved": round(a, 1), "peak": round(l1_peak_gacc, 1), "unit": "Gaccess/s", "frac": round(a / l1_peak_gacc, 4),
                    "line_accesses_per_launch": int(pmc_k["TCP_TOTAL_CACHE_ACCESSES_sum"])}
        if pmc_k.get("GRBM_GUI_ACTIVE") and "avg_us_l1" in pmc_k:
            # share of the L1s' cycles spent stalled, under the counters' own launch time (256 TCPs; GRBM_GUI_ACTIVE sums 8 XCDs)
            cyc = pmc_k["GRBM_GUI_ACTIVE"] / 8.0 * 256.0
            lv["l1"]["stalled_waiting_for_l2"] = round(pmc_k.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / cyc, 3)
            lv["l1"]["stalled_on_tag_conflicts"] = round(pmc_k.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0.0) / cyc, 3)
    if "TCP_TCC_READ_REQ_sum" in pmc_k:
        b = pmc_k["TCP_TCC_READ_REQ_sum"] * 128.0 + pmc_k.get("TCP_TCC_WRITE_REQ_sum", 0.0) * 64.0
        lv["l2"] = {"achieved": round(b / s / 1e9, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(b / s / 1e9 / L2_PEAK_GBS, 4), "bytes_per_launch": int(b),
                    "hit_rate": round(pmc_k["TCC_HIT_sum"] / (pmc_k["TCC_HIT_sum"] + pmc_k["TCC_MISS_sum"]), 3) if (pmc_k.get("TCC_HIT_sum", 0) + pmc_k.get("TCC_MISS_sum", 0)) > 0 else None}
    if "TCC_EA0_RDREQ_sum" in pmc_k:
        rd32 = pmc_k.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        rd = (pmc_k["TCC_EA0_RDREQ_sum"] - rd32) * 128.0 + rd32 * 32.0
        wr64 = pmc_k.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        wr = wr64 * 64.0 + (pmc_k.get("TCC_EA0_WRREQ_sum", 0.0) - wr64) * 32.0
        b = rd + wr
        lv["fabric"] = {"achieved": round(b / s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / s / 1e9 / HBM_PEAK_GBS, 4),
                        "bytes_per_launch": int(b), "read_bytes": int(rd), "write_bytes": int(wr),
                        # what rocprofv3's derived FETCH_SIZE / WRITE_SIZE would print for the same counters (KiB; FETCH_SIZE tallies
                        # 128-byte requests at 64: the x2 of MI355X_MICROARCH.md section HBM, confirmed by the calibration streams)
                        "FETCH_SIZE_KiB_equiv": round(((pmc_k["TCC_EA0_RDREQ_sum"] - rd32) * 64.0 + rd32 * 32.0) / 1024.0, 1), "WRITE_SIZE_KiB_equiv": round(wr / 1024.0, 1),
                        "note": "L2 misses: served by the Infinity Cache or HBM (the TCC_EA0 counters cannot tell them apart, calibration.json); an upper bound of the HBM bytes"}
    return lv


def expected_scaling(workload, world):
    """What N ranks can reach before the transport costs anything, under the three definitions of the line: read from the newest
    profiles/r*/shard_share_probe_<workload>.txt (tools/shard_share_probe.py: one-GPU probes of the last rank's share), not a table
    in this file.  Returns ((value, two in flight, pipelined) or None, the file's path)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"shard_share_probe_{workload}.txt")),
                   key=lambda f: int(re.search(r"profiles[/\\]r(\d+)", f).group(1)))
    if not files:
        return None, "no profiles/r*/shard_share_probe_%s.txt" % workload
    rel = os.path.relpath(files[-1], ROOT)
    for line in open(files[-1]):
        m = re.match(r"\s*1/(\d+)\s", line)
        if m and int(m.group(1)) == world:
            x = [float(v) for v in re.findall(r"\(x\s*([0-9.]+)\)", line)]
            if len(x) == 3:
                return tuple(x), rel
    return None, rel


def frame_hash(img):
    """64 bits of SHA-256 over the bytes of a frame: frames are a pure function of (scene, options, frame index) whatever N is, so the
    hash of the N-rank job's display frame must equal the N = 1 line's."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()[:16]


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
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

    # The exchange of a pixel-sharded job.  torch.distributed stays what starts the ranks, carries the RCCL id to them and reduces
    # the timings; the frames themselves go through the C ABI of include/trhip_comm.h when RCCL is the transport.
    exchange, exchange_name = None, "none"
    if world > 1 and args.shard == "pixels":
        exchange_name = "torch.distributed (%s)" % args.dist_backend
        if args.exchange == "ipc":
            pass      # created below, once the frame batch is known
        elif args.exchange == "native" or (args.exchange == "auto" and args.dist_backend == "nccl"):
            # creating the communicator is collective: first make sure every rank can take part (a rank that cannot load the
            # library must not leave the others waiting inside ncclCommInitRank), then every rank or none
            TC, err = None, None
            try:
                from tauray_amd import comm as TC
                TC.lib()
            except Exception as e:
                err = str(e)
            errs = [None] * world
            dist.all_gather_object(errs, err)
            if any(errs):
                if args.exchange == "native":
                    raise RuntimeError("native exchange unavailable: " + "; ".join(f"rank {i}: {e}" for i, e in enumerate(errs) if e))
                if rank == 0:
                    print("[bench] native exchange unavailable (" + "; ".join(e for e in errs if e) + "): torch.distributed carries the frames", file=sys.stderr)
            else:
                box = [TC.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                exchange = TC.NativeExchange(TC.Comm(local_rank, world, rank, box[0]))
                exchange_name = "libtrhip_comm.so: trhip_gather_partials (grouped ncclSend / ncclRecv, RCCL)"

    W, H = args.width, args.height
    scene = scenes.WORKLOADS[args.workload](W, H)
    if args.views > 1:      # light-field grid (src/tauray.cc:680-727): spacing 0.02, recentering distance 5
        from tauray_amd.scene import generate_camera_grid
        gw = 9 if args.views == 45 else args.views
        scene.cameras = generate_camera_grid(scene.cameras[0], gw, args.views // gw, 0.02, 0.02, 5.0)
    ctx = R.Context(local_rank)
    opt = stage_options(args, scene, R)
    strips = world > 1 and args.shard == "pixels" and args.strategy != "scanline"
    strategy = DISTRIBUTION_SHUFFLED_STRIPS if strips else DISTRIBUTION_SCANLINE
    steps = max(args.steps, MIN_TIMED_FRAMES)           # frames of every timed region
    # Pipelined region.  A rank of a pixel-sharded job traces 1 / N of a frame per launch: at N = 8 a launch no longer fills the GPU,
    # and several frames per launch cost 12 % less per frame (tools/shard_share_probe.py).  The frames are the
    # same frames (tests/test_gpu_parity.py::test_frame_batches_render_the_frames_of_separate_calls).
    B = args.frames_per_launch
    if B <= 0:
        if world > 1 and args.shard == "pixels" and args.views == 1:
            B = 5
        else:       # whole frames: two per launch are worth 2 % on sponza_teapots and 5 % on test.glb, more are not
            B = 2 if (world == 1 and args.views == 1 and args.spp == 1) else 1
    steps_pipelined = ((steps + B - 1) // B) * B
    if world > 1 and args.shard == "pixels" and args.exchange == "ipc":
        # The copy-engine exchange (include/trhip_comm.h trhip_ipc_*): partial frames are written into the display rank's IPC-mapped
        # arena by hipMemcpyAsync - no RCCL kernel on either device - and ordered by tags.  A/B against the RCCL exchange: the display
        # rank's phases below say whether RCCL's kernels wait behind the persistent trace kernels.
        from tauray_amd import comm as TC

        def allgather(blob):
            out = [None] * world
            dist.all_gather_object(out, blob)
            return out
        exchange = TC.IpcExchange(TC.Ipc(local_rank, world, rank, W * H * 16 * max(B, 1) * args.views, args.frames_in_flight, allgather))
        exchange_name = "libtrhip_comm.so: trhip_ipc_gather_partials (hipMemcpyAsync into IPC-mapped memory of the display rank, tags; no RCCL kernels)"
    rr = R.RtRenderer(ctx, scene, opt, (W, H), strategy=strategy, rank=rank, world_size=world,
                      viewports=args.views, shard=args.shard, frames_in_flight=args.frames_in_flight, frames_per_launch=B, exchange=exchange)
    # The renderer of a caller that waits for every frame: no frame slots, one frame per launch; each frame runs as four concurrent
    # lanes (the stage's automatic schedule for frames of >= 1.5 M paths).
    lone = R.RtRenderer(ctx, scene, opt, (W, H), strategy=strategy, rank=rank, world_size=world, viewports=args.views, shard=args.shard,
                        frames_in_flight=1, frames_per_launch=1, exchange=exchange)

    # which kernels shade the frames: asked of the stage (trhip_pt_get_program), and - N > 1 - the same on every rank or nobody renders
    program = lone.check_same_program() if dist is not None else lone.program()
    if rr.program()["identity"] != program["identity"]:
        raise RuntimeError("the two renderers of this process resolved different shading programs")

    def sync_all(r):
        r.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def run_frames(r, n, one_at_a_time=False, times=None):
        per = r.frames_per_launch
        for _ in range((n + per - 1) // per):      # `per` frames per render(); untimed regions round up
            t1 = time.perf_counter() if times is not None else 0.0
            r.reset_accumulation()     # offline frames: accumulation reset, sample counter kept (src/tauray.cc:1101)
            r.render()
            if one_at_a_time:
                r.sync()
                if times is not None:
                    times.append((time.perf_counter() - t1) * 1e3)

    def total_rays(r, elapsed):
        c = r.counters()
        if c["stack_overflows"]:
            raise RuntimeError("BVH traversal stack overflow: results invalid")
        rays = c["closest_rays"] + c["shadow_rays"]
        if dist is not None:
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            q = torch.tensor([rays], dtype=torch.int64, device=f"cuda:{local_rank}")
            dist.all_reduce(q, op=dist.ReduceOp.SUM)
            rays = int(q.item())
        return rays, elapsed, c

    # ---- untimed frames: clocks, page faults and (N > 1) the load balancer
    rr.set_profiling(False, False)
    lone.set_profiling(False, False)
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
        run_frames(rr, max(args.prewarm - every * rounds, 8))
        for _ in range(rounds):
            rr.sync()
            t1 = time.perf_counter()
            run_frames(rr, every)
            rr.sync()
            times = [0.0] * world
            dist.all_gather_object(times, (time.perf_counter() - t1) / (((every + B - 1) // B) * B) * 1e3)
            rr.set_device_workloads(lb.update(times))
        rr.exchange = exchange
        lone.set_device_workloads(list(lb.workloads))
        balance = {"updates": rounds, "frames_per_update": every, "workloads": [round(w, 4) for w in lb.workloads], "workloads_exact": list(lb.workloads),
                   "ms_per_frame_running_free": [round(t, 4) for t in times]}
        sync_all(rr)
        run_frames(rr, args.frames_in_flight * 2)
    else:
        run_frames(rr, args.prewarm)
    sync_all(rr)

    # ---- timed region (`value`): W warm-up frames, then max(K, 50) frames, one at a time, host sync after each, between barriers
    lone.reset_accumulation(reset_sample_counter=True)
    run_frames(lone, args.warmup, True)
    sync_all(lone)
    lone.reset_counters()
    frame_times = []
    t0 = time.perf_counter()
    run_frames(lone, steps, True, frame_times)
    sync_all(lone)
    elapsed = time.perf_counter() - t0
    rays_total, elapsed, _ = total_rays(lone, elapsed)
    lone_lanes, lone_pipes = lone.slots[0].pt.lane_pipes()      # the schedule the timed frames ran with (trhip_pt_get_lane_pipes)
    ms_per_step = elapsed / steps * 1e3
    frame_times.sort()

    # ---- the same frames pipelined (`value_pipelined`): frame slots, B frames per launch, one synchronisation at the end
    rr.reset_accumulation(reset_sample_counter=True)
    run_frames(rr, max(args.warmup, B))
    sync_all(rr)
    rr.reset_counters()
    t0 = time.perf_counter()
    run_frames(rr, steps_pipelined)
    sync_all(rr)
    elapsed_p = time.perf_counter() - t0
    rays_p, elapsed_p, _ = total_rays(rr, elapsed_p)

    # ---- ... and as the reference itself would run them (`value_two_in_flight`): MAX_FRAMES_IN_FLIGHT = 2 (src/context.hh:26), one frame per
    # launch - what a Tauray user gets from the render loop of src/tauray.cc without asking for anything
    two = R.RtRenderer(ctx, scene, opt, (W, H), strategy=strategy, rank=rank, world_size=world, viewports=args.views, shard=args.shard,
                       frames_in_flight=2, frames_per_launch=1, exchange=exchange)
    two.set_profiling(False, False)
    if balance is not None:
        two.set_device_workloads(list(balance["workloads_exact"]))
    two.reset_accumulation(reset_sample_counter=True)
    run_frames(two, max(args.warmup, 4))
    sync_all(two)
    two.reset_counters()
    t0 = time.perf_counter()
    run_frames(two, steps)
    sync_all(two)
    elapsed_2 = time.perf_counter() - t0
    rays_2, elapsed_2, _ = total_rays(two, elapsed_2)
    two.close()

    # ---- N > 1, pixel shards: where a rank's frame goes (one frame at a time, after the timed regions).  Three loops of the same frames:
    # the whole frame; the frame with nothing travelling (transfer.StandaloneExchange: the display rank stitches standing buffers);
    # the path tracing of the rank's share alone.  Differences: what the stitch and the tonemap add on the display rank, and what a
    # rank waits for the transport (on the display rank: for the last partial frame to arrive - behind RCCL's kernels or the copy
    # engines, --exchange ipc - on the others: for their send to drain).  Every rank reports; expectations from DESIGN.md section 6.
    phases = None
    if world > 1 and args.shard == "pixels":
        from tauray_amd.transfer import StandaloneExchange
        n_ph = max(min(steps, 50), 20)

        def loop_ms(body):
            sync_all(lone)
            t1 = time.perf_counter()
            for _ in range(n_ph):
                lone.reset_accumulation()
                body()
                lone.sync()
            return (time.perf_counter() - t1) / n_ph * 1e3
        full_ms = loop_ms(lone.render)
        lone.exchange = StandaloneExchange()
        free_ms = loop_ms(lone.render)
        lone.exchange = exchange
        trace_ms = loop_ms(lone.render_partial)
        mine = {"path_tracing_ms": round(trace_ms, 4), "stitch_and_tonemap_ms": round(free_ms - trace_ms, 4), "transport_wait_ms": round(full_ms - free_ms, 4),
                "frame_ms": round(full_ms, 4)}
        every = [None] * world
        dist.all_gather_object(every, mine)
        phases = {"frames": n_ph, "display_rank": every[0], "other_ranks_max": {k: max(e[k] for e in every[1:]) for k in mine},
                  "note": "one frame at a time with a host sync; transport_wait = whole frame - the same frame with nothing travelling"}
    # what DESIGN.md section 6 expects of this run from one-GPU probes (a rank's strip one frame at a time / with frames in flight, before the transport)
    # what N ranks can reach before the transport costs anything: the whole frame's time over the time of one rank's share, rendered alone on
    # one GPU under each of the three definitions (tools/shard_share_probe.py, profiles/r5/shard_share_probe_sponza_teapots.txt: balanced strips)
    expected, expected_src = expected_scaling(args.workload, world)

    # who rendered: every rank's device as the library sees it, its pipes and its shading program; what RCCL says about the communicator
    try:
        dev_info = ctx.info()
    except AttributeError:      # an older library selected with TRHIP_LIB for an A/B (tools/ab_libs.sh): no trhip_device_get_info
        dev_info = {"hip_device": local_rank, "pci_bus_id": "", "uuid": "", "name": "", "pipe_classes": -1, "hw_queues_env": int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))}
    me = {"rank": rank, "hip_device": dev_info["hip_device"], "pci_bus_id": dev_info["pci_bus_id"], "uuid": dev_info["uuid"], "arch": dev_info["name"],
          "pipe_classes_reachable": dev_info["pipe_classes"], "hw_queues_env": dev_info["hw_queues_env"],
          "lanes": lone_lanes, "lane_pipe_classes": lone_pipes, "program_identity": "%016x" % program["identity"],
          "library_build_id": ("%016x" % R._lib.lib().trhip_build_id()) if hasattr(R._lib.lib(), "trhip_build_id") else None, "pid": os.getpid()}
    comm_info = None
    if exchange is not None and hasattr(getattr(exchange, "comm", None), "info"):
        comm_info = exchange.comm.info()
        me["rccl"] = comm_info
    ranks_info = [me]
    if dist is not None:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)

    result = {
        "metric": "Mray/s (closest-hit + shadow rays traced) @%dx%d, %d bounces, %d spp" % (W, H, args.bounces, args.spp),
        "value": round(rays_total / elapsed / 1e6, 2), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if args.workload != "test_glb" else "reference fixture test/test.glb (81 364 triangles)",
        "steps_effective": steps,
        "value_definition": "SURVEY.md 8(d): one frame at a time, host wall time from before render() to after the stream sync, "
                            f"{steps} frames between barriers (--steps {args.steps}, at least {MIN_TIMED_FRAMES}); frames in flight: see value_pipelined",
        "frame_ms": {"mean": round(sum(frame_times) / len(frame_times), 4), "p50": round(frame_times[len(frame_times) // 2], 4),
                     "min": round(frame_times[0], 4), "max": round(frame_times[-1], 4), "frames": len(frame_times)},
        "value_two_in_flight": round(rays_2 / elapsed_2 / 1e6, 2),
        "two_in_flight": {"ms_per_frame": round(elapsed_2 / steps * 1e3, 4), "frames": steps, "frames_in_flight": 2, "frames_per_launch": 1, "unit": "Mray/s",
                          "definition": "the reference's own MAX_FRAMES_IN_FLIGHT = 2 (src/context.hh:26), one frame per launch, one synchronisation at the end"},
        "value_pipelined": round(rays_p / elapsed_p / 1e6, 2),
        "pipelined": {"ms_per_frame": round(elapsed_p / steps_pipelined * 1e3, 4), "frames": steps_pipelined, "frames_in_flight": args.frames_in_flight,
                      "frames_per_launch": B, "unit": "Mray/s"},
        "config": {"workload": args.workload, "triangles": scene.triangle_count, "width": W, "height": H, "bounces": args.bounces,
                   "spp": args.spp, "sampler": ["uniform-random", "sobol-owen", "sobol-z2", "sobol-z3"][opt.sampler],
                   "preset": args.preset, "film": ["point", "box", "blackman-harris"][opt.film], "regularization": round(opt.regularization_gamma, 3),
                   "tri_light_mode": ["area", "solid-angle", "hybrid"][opt.tri_light_mode],
                   # what the library launched (trhip_pt_get_program), not what the options suggest
                   "shading_program": {"general": "general kernels (or unknown: see identity)", "cli": "command-line set, ahead of time", "compiled": "compiled for the option set (hipRTC / kernel cache)"}[program["kind"]]
                                      + (", IEEE fp32" if program["ieee"] else ", Vulkan-grade arithmetic"),
                   "shading_program_identity": "%016x" % program["identity"],
                   # lanes of a timed frame and the hardware pipe of each lane's stream: lanes on one pipe would run one after the other
                   "lanes": lone_lanes, "lane_pipe_classes": lone_pipes,
                   "parallelism": ({"pixels": ("shuffled strips x%d, balanced shares + RCCL gather" if balance else "shuffled strips x%d + RCCL gather") if strips
                                    else "scanline-sharded x%d + RCCL gather", "views": "view-sharded x%d, no exchange",
                                    "samples": "sample-sharded x%d + RCCL reduce"}[args.shard] % world) if world > 1 else "single GPU",
                   "views": args.views, "frames_in_flight": 1, "frames_per_launch": 1, "prewarm_frames": args.prewarm, "exchange": exchange_name,
                   "scene_hash": scenes.scene_hash(scene)},
        **({"rank_phases": phases} if phases else {}),
        "ranks": ranks_info,
        "devices_distinct": len({(r["pci_bus_id"], r["uuid"]) for r in ranks_info}),
        **({"rccl": {"nranks": comm_info["nranks"], "version": comm_info["rccl_version"], "nranks_seen_by_every_rank": sorted({r.get("rccl", {}).get("nranks") for r in ranks_info}),
                     "source": "trhip_comm_get_info: ncclCommCount / ncclGetVersion on the communicator the frames travel through"}} if comm_info else {}),
        **({"scaling_expected_vs_one_gpu": {"value": expected[0], "value_two_in_flight": expected[1], "value_pipelined": expected[2],
                                             "source": expected_src + ": one-GPU probes of a rank's share, before the transport (DESIGN.md section 6)"}}
           if (expected and world > 1) else {}),
        "accel_build_ms": round(rr.scene_update.accel["build_ms"], 2),
        **({"load_balance": {k: v for k, v in balance.items() if k != "workloads_exact"}} if balance else {}),
        "rays_per_frame": rays_total // steps,
        "msample_per_s": round(W * H * args.views * args.spp * steps / elapsed / 1e6, 2),
    }

    if world == 1 and args.sustained_frames > 0:      # a timed region long enough for a 1 Hz utilisation sampler to see
        sync_all(rr)
        t1 = time.perf_counter()
        run_frames(rr, args.sustained_frames)
        sync_all(rr)
        dt = time.perf_counter() - t1
        n_s = ((args.sustained_frames + B - 1) // B) * B
        result["sustained"] = {"frames": n_s, "ms_per_frame": round(dt / n_s * 1e3, 4), "value": round(result["rays_per_frame"] * n_s / dt / 1e6, 2),
                               "unit": "Mray/s", "mode": "pipelined"}

    # ---- roofline of the dominant kernel (k_trace_closest), rank 0
    if not args.no_roofline:
        # every rank takes part (a frame of a multi-GPU job ends in an exchange between the ranks); rank 0's kernels are reported.
        # re-run of the identical frames (same frame indices) with per-kernel HIP events; detailed timing serialises the
        # frame (one lane, no shadow/closest overlap), so every kernel is measured owning the chip (instance k_trace_closest<false, true>)
        n_r = ((min(steps, 20) + B - 1) // B) * B
        rr.set_profiling(False, True)
        rr.reset_accumulation(reset_sample_counter=True)
        run_frames(rr, max(args.warmup, B), True)
        rr.reset_counters()
        run_frames(rr, n_r, True)
        timings = rr.timings()
        launches = max(timings["trace_closest_launches"], 1)
        avg_ms = timings["trace_closest_ms"] / launches
        # counted re-run for the algorithmic byte model and the rays per wave-level instruction
        rr.set_profiling(True, False)
        rr.reset_accumulation(reset_sample_counter=True)
        run_frames(rr, max(args.warmup, B))
        rr.reset_counters()
        run_frames(rr, n_r)
        c = rr.counters()
        ph = rr.phase_counters()
        rr.set_profiling(False, False)
        # bytes per SURVEY.md section 8(d), restricted to what trace kernels touch; the closest-hit kernel's share of
        # node/triangle work is apportioned by ray count (both trace kernels walk the same structure)
        closest_share = c["closest_rays"] / max(c["closest_rays"] + c["shadow_rays"], 1)
        node_bytes = rr.scene_update.accel["node_bytes"]   # what one node visit reads (trhip_accel_info)
        trace_bytes = (c["node_visits"] * node_bytes + c["tri_tests"] * 48 + c["alpha_tests"] * 52) * closest_share \
            + c["closest_rays"] * (16 + 16 + 16 + 16)   # ray origin + direction + misc read, hit record write
        bytes_per_launch = trace_bytes / launches
        algorithmic = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        frame_bytes = (c["node_visits"] * node_bytes + c["tri_tests"] * 48 + c["alpha_tests"] * 52 + c["surface_hits"] * 268
                       + (c["closest_rays"] + c["shadow_rays"]) * (2 * 48 + 2 * 20) + W * H * args.views * args.spp * 16 * n_r) / n_r
        kernel_avg_ms = {k: (timings[v + "_ms"] / max(timings[v + "_launches"], 1)) for k, v in HOT_KERNELS.items()}
        roof = {
            "kernel": "k_trace_closest", "avg_launch_ms": round(avg_ms, 4), "launches": int(launches), "frames_per_launch": B,
            "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None, "levels": {},
            "algorithmic_GBps": round(algorithmic, 1), "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "algorithmic_note": "SURVEY.md 8(d) counted-work bytes / launch time: node visits x 112 B etc.; served by L1 / L2 / Infinity Cache, "
                                "so it may exceed the HBM peak and is not a roofline fraction",
            "frame_algorithmic_GBps": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            # what a frame cannot avoid moving even with a perfectly cached tree: ray + hit records written and read once, one
            # framebuffer write
            "frame_compulsory_GBps": round(((c["closest_rays"] + c["shadow_rays"]) * (2 * 48 + 2 * 20) + W * H * args.views * args.spp * 16 * n_r)
                                           / n_r / (ms_per_step * 1e-3) / 1e9, 1),
            "kernel_ms_per_frame": {k: round(timings[k + "_ms"] / n_r, 4) for k in ("trace_closest", "trace_shadow", "shade", "raygen", "resolve")},
            "node_visits_per_ray": round(c["node_visits"] / max(c["closest_rays"] + c["shadow_rays"], 1), 2),
            "tri_tests_per_ray": round(c["tri_tests"] / max(c["closest_rays"] + c["shadow_rays"], 1), 2),
        }
        if rank == 0:
            try:
                valu_peak = ctx.calibrate_valu()
            except Exception as e:      # an older libtrhip.so
                valu_peak = None
                roof["valu_calibration_error"] = str(e)
            roof["valu_peak_measured_ginst_per_s"] = round(valu_peak, 1) if valu_peak else None
            try:
                l1_peak = ctx.calibrate_l1()
            except Exception as e:
                l1_peak = None
                roof["l1_calibration_error"] = str(e)
            roof["l1_peak_measured_gaccess_per_s"] = round(l1_peak, 1) if l1_peak else None
            if world > 1:       # a shard's launches are not the launches of the N = 1 line: the counter passes belong to that line
                pmc, pmc_err = {}, "N > 1: the counter passes (VALU / L2 / fabric levels) are part of the N = 1 line"
            else:
                pmc, pmc_err = ({}, "switched off (--no-pmc)") if args.no_pmc else run_pmc_passes(args, B, args.pmc_dump)
            if pmc_err:
                roof["pmc_error"] = pmc_err
            lv = level_fractions(pmc.get("k_trace_closest"), avg_ms, valu_peak, l1_peak)
            # rays per wave-level instruction of the traversal: node visits of the closest-hit rays / node phases (a phase = one pass of
            # a wave over the node code, one ray per lane or one ray per quad); the hardware lane count (hw_lanes_per_inst) counts a
            # quad's four lanes as four
            phases = ph["lane_node_phases"] + ph["quad_node_phases"]
            if "valu" in lv and phases > 0:
                rpi = ph["closest_node_visits"] / phases
                lv["valu"]["rays_per_node_phase"] = round(rpi, 2)
                lv["valu"]["useful_frac"] = round(lv["valu"]["frac"] * rpi / 64.0, 4)
                lv["valu"]["node_phases"] = {"one_ray_per_lane": ph["lane_node_phases"], "one_ray_per_quad": ph["quad_node_phases"],
                                             "per_lane_by_live_rays_1_8_to_57_64": ph["lane_node_phase_hist"]}
            roof["levels"] = lv
            if lv:
                b = max(lv, key=lambda k: lv[k]["frac"])
                roof.update({"bound": b, "achieved": lv[b]["achieved"], "peak": lv[b]["peak"], "unit": lv[b]["unit"], "frac": lv[b]["frac"]})
                if "fabric" in lv:
                    roof["traffic"] = lv["fabric"]["bytes_per_launch"]
                    roof["traffic_note"] = "L2-miss bytes per launch (read requests x 128 B + write requests x 64 / 32 B; counters under rocprofv3 in this run)"
                if b == "l1":
                    roof["bound_note"] = ("the busiest unit is the vector L1 of every CU: a traversal step reads its 128-byte node with seven 16-byte loads per "
                                          "lane, seven line accesses at one per clock and CU; the rest of the L1s' cycles go to waiting for L2 data "
                                          "(levels.l1.stalled_waiting_for_l2) - the kernel is a chain of dependent fetches at six waves per SIMD")
                else:
                    roof["bound_note"] = ("no level is near its peak: the kernel waits on dependent node fetches (wait_fraction) at %d waves per SIMD"
                                          % 6) if lv[b]["frac"] < 0.6 else None
            roof["other_kernels"] = {k: {"avg_launch_ms": round(kernel_avg_ms[k], 4), "levels": {n: {"frac": d["frac"], "achieved": d["achieved"], "unit": d["unit"]}
                                                                                                  for n, d in level_fractions(pmc.get(k), kernel_avg_ms[k], valu_peak, l1_peak).items()}}
                                     for k in ("k_trace_shadow", "k_shade")}
            # the frames `value` and `value_pipelined` time (weak #3 / #4 of round 4's review: the roofline kernel is a proxy that is only
            # launched under detailed timing): how busy the same counters say the chip is over those frames
            if pmc:
                roof["timed_frames"] = {"one_frame_at_a_time": whole_frame_utilisation(pmc, ms_per_step, valu_peak, l1_peak),
                                        "pipelined": whole_frame_utilisation(pmc, elapsed_p / steps_pipelined * 1e3, valu_peak, l1_peak),
                                        "note": "counts (vector instructions, L1 line accesses, L2-miss bytes) of every hot-kernel launch of a frame, from the serialised counter "
                                                "passes, over the frame's wall time and the unit's peak; the kernels of a timed frame overlap on four streams"}
            roof["pmc_source"] = "rocprofv3 --pmc passes run by this process (bench.py --pmc-child); factors: profiles/r3/calibration.json, l1_tag_rate.json"
        result["roofline"] = roof

    # ---- CPU baseline: the oracle (a port; the reference has no CPU path) on the host cores, rank 0, N = 1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as OB
        osc = OB.OracleScene(scene)
        oopt = OB.options_for_scene(scene, **args.option_kw)
        # threads: what this process may actually run on - its affinity mask, bounded by the cgroup's CPU quota - not os.cpu_count()
        affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        except (OSError, ValueError):
            try:
                q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    quota = q / per
            except (OSError, ValueError):
                pass
        cores = max(1, min(affinity, int(quota + 0.5)) if quota else affinity)
        # one thread first, on a quarter-size frame of the same camera (the full frame would take a minute): the rate the threads scale from
        w1, h1 = max(W // 4, 16), max(H // 4, 16)
        osc.reset_counters()
        t0 = time.perf_counter()
        osc.render_pt(oopt, w1, h1, frame_counter=0, threads=1, viewports=args.views)
        dt1 = time.perf_counter() - t0
        oc1 = osc.counters()
        one_thread = (oc1["closest_rays"] + oc1["shadow_rays"]) / dt1 / 1e6
        osc.reset_counters()
        frames = 0
        t0 = time.perf_counter()
        ref0 = None
        while True:
            img = osc.render_pt(oopt, W, H, frame_counter=frames, threads=cores, viewports=args.views)
            if frames == 0:
                ref0 = img      # frame 0 of the workload as the oracle renders it: what the HIP frame of the same index is held against below
            frames += 1
            if time.perf_counter() - t0 >= args.cpu_seconds:
                break
        dt = time.perf_counter() - t0
        oc = osc.counters()
        import shutil
        vk_icd = any(os.path.isdir(d) and os.listdir(d) for d in ("/usr/share/vulkan/icd.d", "/etc/vulkan/icd.d"))
        tauray_bin = shutil.which("tauray")
        result["cpu_baseline"] = {
            "value": round((oc["closest_rays"] + oc["shadow_rays"]) / dt / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "threads_used": cores, "affinity_cpus": affinity, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(),
            "mray_per_s_one_thread": round(one_thread, 4), "one_thread_sample": f"one {w1}x{h1} frame, {dt1:.1f} s",
            "scaling_efficiency": round((oc["closest_rays"] + oc["shadow_rays"]) / dt / 1e6 / (one_thread * cores), 3),
            "sample": f"{frames} full {W}x{H} frame(s) of the same workload, {dt:.1f} s of wall time, OpenMP schedule(dynamic) over 16x16 tiles",
            "ms_per_frame": round(dt / frames * 1e3, 1),
            # north_star asks for Tauray's raster fallback on the host cores beside the number: it needs a Vulkan software ICD and a
            # compiled Tauray; neither can be installed here
            "raster_fallback": ("available but not timed by this script" if (vk_icd and tauray_bin)
                                else "unavailable: no %s on this box" % " and no ".join(([] if vk_icd else ["Vulkan ICD"]) + ([] if tauray_bin else ["Tauray binary"]))),
        }

        # ---- parity of the kernels that were just timed: the HIP frame of frame index 0 against the oracle's frame 0 (same scene, camera,
        # options, sampler state), at the full size the bench times.  The oracle is IEEE fp32; the default shading arithmetic is
        # Vulkan-grade (2.5-ulp division, hardware square roots), so single paths may take another branch: the figures are the share of
        # pixels outside 1 % (+ 0.01 absolute), the relative error of the image mean and the RMS difference of the radiance.
        lone.set_profiling(False, False)
        lone.reset_accumulation(reset_sample_counter=True)
        lone.render()
        sync_all(lone)
        hip0 = lone.download("color")[:args.views]
        a, b = hip0[..., :3].astype(np.float64), ref0[..., :3].astype(np.float64)
        finite = bool(np.isfinite(hip0).all())
        rel = np.abs(a - b) / (np.abs(b) + 1e-2)
        outside = float((rel.max(-1) > 1e-2).mean())
        mean_rel = abs(a.mean() - b.mean()) / max(b.mean(), 1e-30)
        rms = float(np.sqrt(np.mean((a - b) ** 2)))
        # The RMS of ONE 1-spp frame is not north_star's "image L2 < 1e-3": that figure is for the converged image (config 3, 4096 spp:
        # RMS 1.3e-5 IEEE / 2.9e-5 Vulkan-grade, tests/test_gpu_parity.py).  A single sample per pixel has no averaging: the handful of
        # paths (0.03 % of the pixels) that take another branch under 2.5-ulp division each differ by a whole path's radiance, and they
        # alone set the RMS (1.3e-3 at the default arithmetic, 5.6e-4 at IEEE fp32 where the same happens on ~0.01 % of the pixels).
        # The bound on it is therefore stated per arithmetic, and the IEEE kernels' figure is printed beside the default's.
        rms_bound = 1.5e-3 if args.ieee_shading else 3e-3
        result["parity"] = {
            "against": "oracle (CPU restatement of the reference's GLSL, IEEE fp32), frame index 0 of this workload at %dx%d" % (W, H),
            "pixels_outside_1e-2": round(outside, 6), "mean_rel_err": float("%.3e" % mean_rel), "rms": float("%.3e" % rms),
            "rms_bound": rms_bound,
            "rms_definition": "sqrt(mean((hip - oracle)^2)) over the RGB of ONE 1-spp frame (no averaging over samples; north_star's 1e-3 is for the "
                              "converged image - config 3 at 4096 spp: 1.3e-5 IEEE, 2.9e-5 Vulkan-grade, asserted in tests/test_gpu_parity.py)",
            "bit_identical_pixels": round(float((hip0[..., :3] == ref0[..., :3]).all(-1).mean()), 4),
            "finite": finite,
            "kernels": "the instances of the timed frames (no counting, no per-kernel timing), one frame with a host sync",
            "shading_arithmetic": "IEEE fp32" if args.ieee_shading else "Vulkan-grade",
            "pass": bool(finite and outside < 5e-3 and mean_rel < 2e-3 and rms < rms_bound),
        }
        if not args.ieee_shading:      # the same frame from the IEEE kernels (trhip_pt_set_shading_arithmetic 1), not timed
            ieee = R.RtRenderer(ctx, scene, opt, (W, H), strategy=strategy, rank=rank, world_size=world, viewports=args.views, shard=args.shard,
                                frames_in_flight=1, frames_per_launch=1, exchange=exchange)
            for slot in ieee.slots:
                slot.pt.set_shading_arithmetic(True)
            ieee.set_profiling(False, False)
            ieee.reset_accumulation(reset_sample_counter=True)
            ieee.render()
            sync_all(ieee)
            hi = ieee.download("color")[:args.views]
            ieee.close()
            ai = hi[..., :3].astype(np.float64)
            rel_i = np.abs(ai - b) / (np.abs(b) + 1e-2)
            result["parity"]["ieee_kernels"] = {
                "pixels_outside_1e-2": round(float((rel_i.max(-1) > 1e-2).mean()), 6),
                "mean_rel_err": float("%.3e" % (abs(ai.mean() - b.mean()) / max(b.mean(), 1e-30))),
                "rms": float("%.3e" % float(np.sqrt(np.mean((ai - b) ** 2)))), "rms_bound": 1.5e-3,
                "bit_identical_pixels": round(float((hi[..., :3] == ref0[..., :3]).all(-1).mean()), 4)}
            result["parity"]["pass"] = bool(result["parity"]["pass"] and result["parity"]["ieee_kernels"]["rms"] < 1.5e-3)
        if not result["parity"]["pass"]:
            print("bench.py: the HIP frame deviates from the oracle's: %s" % json.dumps(result["parity"]), file=sys.stderr)

    # frame 0 again on every rank, outside all timing: the display frame's hash is part of every line.  Frames do not depend on N (RNG keyed by
    # absolute pixel, shards stitched bit for bit), so the N-rank line's hash equals the N = 1 line's of the same workload and options.
    lone.set_profiling(False, False)
    lone.reset_accumulation(reset_sample_counter=True)
    lone.render()
    sync_all(lone)
    if rank == 0:
        disp = lone.download("display")[:args.views]
        result["display_frame"] = {"frame_index": 0, "sha256_64": frame_hash(disp), "shape": list(disp.shape),
                                   "note": "tonemapped frame 0 on the display rank; identical for every N by construction (tests/test_multi_rank_gloo.py, test_multi_device.py)"}
        if args.save_display:
            np.save(args.save_display, disp)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
