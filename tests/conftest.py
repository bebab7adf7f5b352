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


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def ensure_built():
    """Build libnfft_hip.so (hipcc cross-compiles without a GPU) and the C oracle if they are missing or stale."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_nfft_hip_build", os.path.join(ROOT, "torch_nfft_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)
    from oracle import ndft_cpu
    ndft_cpu.build()


def pytest_sessionstart(session):
    ensure_built()
