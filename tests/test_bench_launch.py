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
