#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<round>/.

usage: python tools/profile_summary.py gpurun_out/prof_<tag> profiles/<round>
Copies the rocprofv3 --stats kernel table and the bench line of the same run per workload, and writes
pmc_summary.json: per hot kernel the per-launch averages of every counter pass plus derived figures
(VALU lane utilisation, VALU issue busy, HBM traffic range per MI355X_MICROARCH.md's FETCH_SIZE caveat).
"""
import collections, csv, glob, json, os, re, shutil, sys

CLOCK_HZ = 2.4e9      # MI355X peak engine clock
SIMDS = 256 * 4


def per_kernel(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in rows:
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: dict(launches=len(seen[k]), avg_us=dur[k] / len(seen[k]) / 1e3,
                    **{c: v / len(seen[k]) for c, v in agg[k].items()}) for k in agg}


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    summary = {}
    for wdir in sorted(glob.glob(os.path.join(src, "*"))):
        w = os.path.basename(wdir)
        stats = glob.glob(os.path.join(wdir, "stats", "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            shutil.copy(stats[0], os.path.join(dst, f"{w}_kernel_stats.csv"))
        bj = os.path.join(wdir, "bench_under_rocprof.json")
        if os.path.exists(bj):
            lines = [l for l in open(bj) if l.startswith("{")]
            if lines:
                open(os.path.join(dst, f"{w}_bench_under_rocprof.json"), "w").write(lines[-1])
        out = collections.defaultdict(dict)
        for p in ("fetch", "write", "sq", "ta"):
            cc = glob.glob(os.path.join(wdir, p, "**", "*counter_collection.csv"), recursive=True)
            if not cc:
                continue
            for k, v in per_kernel(cc[0]).items():
                if not ("trace" in k or "shade" in k):
                    continue
                out[k][f"launches_{p}"] = v.pop("launches")
                out[k][f"avg_us_{p}"] = round(v.pop("avg_us"), 1)
                for c, x in v.items():
                    out[k][c + "_per_launch"] = round(x, 1)
        for k, o in out.items():
            if "SQ_INSTS_VALU_per_launch" in o and o["SQ_INSTS_VALU_per_launch"] > 0:
                o["valu_lane_utilisation"] = round(o["SQ_THREAD_CYCLES_VALU_per_launch"] / o["SQ_INSTS_VALU_per_launch"] / 64.0, 3)
                o["valu_issue_busy"] = round(o["SQ_INSTS_VALU_per_launch"] * 4 / (o["avg_us_sq"] * 1e-6 * CLOCK_HZ * SIMDS), 3)
                o["wait_fraction"] = round(o["SQ_WAIT_ANY_per_launch"] / o["SQ_WAVE_CYCLES_per_launch"], 3)
            if "FETCH_SIZE_per_launch" in o:   # KiB; under-reports wide coalesced reads by up to 2x on gfx950
                lo = o["FETCH_SIZE_per_launch"] * 1024
                wr = o.get("WRITE_SIZE_per_launch", 0.0) * 1024
                o["hbm_traffic_bytes_per_launch_range"] = [int(lo + wr), int(2 * lo + wr)]
        summary[w] = out
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({w: {k: {x: o[x] for x in ("valu_lane_utilisation", "valu_issue_busy", "wait_fraction", "hbm_traffic_bytes_per_launch_range") if x in o}
                          for k, o in v.items()} for w, v in summary.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
