#!/usr/bin/env python3
"""Markdown table of the option-set bench lines of one profiling session (tools/profile_r4.sh):
    python tools/debug/option_sets_table.py profiles/r4 > profiles/r4/option_sets.md
Every row is one `python bench.py <flags> --steps 20 --warmup 5` line on sponza_teapots; the reference row is the command-line set
(`sponza_teapots_bench.json`, the line the driver measures)."""
import glob
import json
import os
import sys


def line(path):
    rows = [x for x in open(path) if x.startswith("{")]
    return json.loads(rows[-1]) if rows else None


def main(d):
    base = line(os.path.join(d, "sponza_teapots_bench.json"))
    out = ["# Option sets on sponza_teapots (1920x1080, 1 spp), one box, one session", "",
           "`flags` are `bench.py`'s; `frame` = one frame at a time (`value`'s definition), `pipelined` = four frames in flight; `vs default` compares",
           "the frame with the command-line set's of the same session.  Presets keep their own bounce count (quality 4, accumulation 5, reference 8),",
           "so their frames are compared per ray (Mray/s).", "",
           "| flags | shading program | bounces | frame ms | Mray/s | pipelined ms | Mray/s | frame vs default | Mray/s vs default |",
           "|---|---|---|---|---|---|---|---|---|"]

    def row(name, r):
        c = r["config"]
        out.append("| `%s` | %s | %d | %.3f | %.0f | %.3f | %.0f | %+.1f %% | %+.1f %% |" % (
            name, c.get("shading_program", ""), c.get("bounces", 0), r["ms_per_step"], r["value"], r["pipelined"]["ms_per_frame"],
            r["value_pipelined"], 100 * (r["ms_per_step"] / base["ms_per_step"] - 1), 100 * (r["value"] / base["value"] - 1)))

    row("(none: the command-line set)", base)
    for f in sorted(glob.glob(os.path.join(d, "option_set_*.json"))):
        r = line(f)
        if not r or "value" not in r:
            continue
        n = os.path.basename(f)[len("option_set_"):-len(".json")]
        flags = " ".join("--" + t for t in n.replace("sampler_", "sampler ").replace("preset_", "preset ").replace("_with_counters", "").split("_"))
        if n.endswith("_with_counters"):
            flags += " (with the counter passes)"
        row(flags, r)
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r4")
