import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


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
