#!/usr/bin/env python3
"""Condenses gpurun_out/<cal>/ (tools/calibrate_box.sh) into profiles/<round>/calibration.json.

usage: python tools/calibration_summary.py gpurun_out/cal profiles/r3/calibration.json

Every configuration of tools/ubench/calibrate runs twice under a counter pass (warm-up + one repetition); the second dispatch
is matched, in launch order, with the configuration's line of the same run's JSON (known instructions / bytes).  The output
holds, per configuration, the counters per dispatch next to the known work, and the factors bench.py uses:

  valu_peak_ginst_per_s      wave-level v_fma_f32 per second at eight waves per SIMD (the VALU roofline's peak)
  bytes_per_TCP_TCC_READ_REQ bytes one L1 -> L2 read request stands for, by access pattern
  bytes_per_TCC_EA0_RDREQ    bytes one L2 -> fabric read request stands for (FETCH_SIZE = requests x 64 B / 1024 under-reports
                             128-byte requests by 2: MI355X_MICROARCH.md section HBM)
"""
import collections
import csv
import glob
import json
import os
import sys


def dispatches(path):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        e = disp.setdefault(d, {"kernel": r["Kernel_Name"], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [e for _, e in sorted(disp.items()) if "k_cal_" in e["kernel"]]


def main(src, dst):
    plain = [json.loads(l) for l in open(os.path.join(src, "plain.jsonl")) if l.startswith("{")]
    device, configs = plain[0], plain[1:]
    out = {"device": device, "configs": []}
    for i, cfg in enumerate(configs):
        out["configs"].append({"plain": cfg, "counters": {}, "us_under_counters": {}})
    for sub in sorted(glob.glob(os.path.join(src, "*", "t_counter_collection.csv"))):
        name = os.path.basename(os.path.dirname(sub))
        d = dispatches(sub)
        if len(d) != 2 * len(configs):
            print(f"pass {name}: {len(d)} calibration dispatches for {len(configs)} configurations, skipped", file=sys.stderr)
            continue
        for i in range(len(configs)):
            e = d[2 * i + 1]
            assert configs[i]["kernel"] in e["kernel"], (configs[i]["kernel"], e["kernel"])
            out["configs"][i]["counters"].update({k: v for k, v in e["c"].items()})
            out["configs"][i]["us_under_counters"][name] = round(e["us"], 1)
    # ---- derived factors
    f = {}
    for c in out["configs"]:
        p, k = c["plain"], c["counters"]
        tag = p["kernel"] + (":" + p["footprint"] if "footprint" in p else "") + (":w%d" % p["waves_per_simd"] if "waves_per_simd" in p else "") \
            + (":%s:%dnodes" % (p["variant"], p["nodes"]) if "variant" in p else "")
        d = {}
        if "wave_insts" in p:
            if k.get("SQ_INSTS_VALU"):
                d["SQ_INSTS_VALU_per_known_inst"] = round(k["SQ_INSTS_VALU"] / p["wave_insts"], 4)
            if k.get("SQ_ACTIVE_INST_VALU") and k.get("SQ_INSTS_VALU"):
                d["SQ_ACTIVE_INST_VALU_per_inst"] = round(k["SQ_ACTIVE_INST_VALU"] / k["SQ_INSTS_VALU"], 4)
            if k.get("SQ_BUSY_CYCLES") and k.get("GRBM_GUI_ACTIVE"):
                d["effective_clock_ghz_under_counters"] = round(k["GRBM_GUI_ACTIVE"] / (c["us_under_counters"].get("sq", 0) * 1e3), 3) if c["us_under_counters"].get("sq") else None
            d["ginst_per_s"] = p["ginst_per_s"]
        else:
            nbytes = p.get("bytes", p.get("bytes_lines_128"))
            if nbytes is None:      # k_cal_gather_var: lines of 128 bytes touched = visits (a node never straddles a line)
                nbytes = p["visits"] * 128.0
                d["Gvisits_per_s"] = p["Gvisits_per_s"]
            for cn in ("TCP_TCC_READ_REQ_sum", "TCC_EA0_RDREQ_sum", "TCC_REQ_sum", "TCC_READ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum"):
                if k.get(cn):
                    d["bytes_per_" + cn] = round(nbytes / k[cn], 2)
            if k.get("FETCH_SIZE"):
                d["bytes_per_FETCH_SIZE_KiB"] = round(nbytes / k["FETCH_SIZE"], 1)
            if k.get("WRITE_SIZE"):
                d["bytes_per_WRITE_SIZE_KiB"] = round(nbytes / k["WRITE_SIZE"], 1)
            if k.get("TCC_HIT_sum") is not None and k.get("TCC_MISS_sum") is not None and (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]) > 0:
                d["l2_hit_rate"] = round(k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]), 4)
            if k.get("TCP_TOTAL_CACHE_ACCESSES_sum") and k.get("TCP_TCC_READ_REQ_sum"):
                d["l1_accesses_per_l2_request"] = round(k["TCP_TOTAL_CACHE_ACCESSES_sum"] / k["TCP_TCC_READ_REQ_sum"], 3)
            d["GBps"] = p.get("GBps", p.get("GBps_112"))
        f[tag] = d
    out["factors_by_config"] = f
    fma8 = [c["plain"] for c in out["configs"] if c["plain"]["kernel"] == "k_cal_fma" and c["plain"].get("waves_per_simd") == 8]
    out["valu_peak_ginst_per_s"] = fma8[0]["ginst_per_s"] if fma8 else None
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps({"valu_peak_ginst_per_s": out["valu_peak_ginst_per_s"], "factors": f}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
