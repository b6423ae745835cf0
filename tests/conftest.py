import importlib.util
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


def load_pkg():
    """Import the package from its (non-identifier) directory name `llama.cpp-omni_amd`."""
    name = "llama_cpp_omni_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "llama.cpp-omni_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def be(pkg):
    """The MI355X backend through its C-ABI.  No fallback: a missing library / GPU is an error, not a skip."""
    return pkg.backend(0)


@pytest.fixture(scope="session")
def ref_be(pkg):
    """The REAL reference CPU backend (oracle/_ref), when it travelled with the snapshot."""
    from oracle.ref_backend import make_ref_cpu_backend, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref/libggml-ref.so not built (needs /root/reference at build time)")
    return make_ref_cpu_backend(pkg, min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden():
    return {n: np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("quant", "ops", "tiny_model")}


def nmse(a, b):
    """The reference's own error measure (tests/test-backend-ops.cpp:225): sum((a-b)^2) / sum(b^2)."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    d = ((a - b) ** 2).sum()
    n = (b ** 2).sum()
    return float(d / n) if n > 0 else float(d)
