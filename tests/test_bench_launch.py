"""`python bench.py --gpus N` has to start its own ranks: the driver calls it without a launcher (no RANK / MASTER_ADDR)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_self_launches_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--one-device",
                        "--rendezvous-only"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"rendezvous": 2, "rank_sum": 3}


def test_bench_defaults_name_the_million_triangle_scene():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.workload == "sponza_teapots" and a.gpus == 1


def test_expected_scaling_is_read_from_the_newest_probe_record(tmp_path):
    """bench.py's `scaling_expected_vs_one_gpu` is not a table in the script: it comes from the newest profiles/r*/shard_share_probe_<workload>.txt
    (tools/shard_share_probe.py), so the figure cannot go stale behind a kernel change."""
    import glob
    import re
    import sys
    sys.path.insert(0, ROOT)
    import bench
    files = glob.glob(os.path.join(ROOT, "profiles", "r*", "shard_share_probe_sponza_teapots.txt"))
    newest = max(files, key=lambda f: int(re.search(r"profiles[/\\]r(\d+)", f).group(1)))
    got, src = bench.expected_scaling("sponza_teapots", 8)
    assert src == os.path.relpath(newest, ROOT)
    row = [l for l in open(newest) if re.match(r"\s*1/8\s", l)][0]
    assert got == tuple(float(v) for v in re.findall(r"\(x\s*([0-9.]+)\)", row)) and len(got) == 3 and 3.0 < got[0] < 8.0
    assert bench.expected_scaling("sponza_teapots", 3)[0] is None          # no such share in the record: no figure, not a guess
    assert bench.expected_scaling("no_such_workload", 8) == (None, "no profiles/r*/shard_share_probe_no_such_workload.txt")
    assert len(bench.frame_hash(__import__("numpy").zeros((2, 2, 4), "float32"))) == 16
