import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cpus():
    """CPUs this process may run on: the affinity mask bounded by the cgroup's quota (a GPU box of this pool shows 256 CPUs and grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


# the oracle's OpenMP loops: as many threads as there are CPUs to run them (256 threads on 16 CPUs cost more than they compute)
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cpus()))
# what tauray_amd/_lib.py sets on import - but _has_gpu() below starts the HIP runtime first, and the runtime reads it once
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    """A device the HIP runtime can see.  Asked of the runtime itself: importing torch for this costs a fresh GPU box a minute or more of
    paging before the first test (the whole -m gpu suite runs 233 s on a fresh box and 143 s on a warm one)."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    try:      # no HIP runtime where the loader looks: ask torch after all
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def device_count():
    """GPUs of this node as the HIP runtime counts them (0 without a runtime or a device): tests/test_multi_device.py needs two."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    return 0


def pytest_ignore_collect(collection_path, config):
    # `-m gpu` (the GPU box): the files that hold CPU tests only are not even imported - tests/test_multi_rank_gloo.py imports torch at
    # module level for its gloo jobs, which is the same minute of paging again
    if (config.option.markexpr or "").strip() == "gpu" and collection_path.name in ("test_multi_rank_gloo.py",):
        return True
    return None


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def test_glb_512():
    from tauray_amd.gltf import load_glb
    return load_glb(os.path.join(GOLDEN, "test.glb"), 512, 512)


@pytest.fixture(scope="session")
def test_glb_128():
    from tauray_amd.gltf import load_glb
    return load_glb(os.path.join(GOLDEN, "test.glb"), 128, 128)


@pytest.fixture(scope="session")
def oracle_scene_512(oracle, test_glb_512):
    return oracle.OracleScene(test_glb_512)


@pytest.fixture(scope="session")
def oracle_scene_128(oracle, test_glb_128):
    return oracle.OracleScene(test_glb_128)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"validate_{name}.npz"))["rgb"].astype(np.float32)
