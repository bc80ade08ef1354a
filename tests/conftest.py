import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the REFERENCE (its Python + its compiled CPU backend), tests/golden/make_golden.py."""
    return np.load(os.path.join(GOLDEN_DIR, "ops_golden.npz"))


@pytest.fixture(scope="session")
def golden_e2e():
    return np.load(os.path.join(GOLDEN_DIR, "minkunet_e2e_golden.npz"))


@pytest.fixture()
def oracle_backend(monkeypatch):
    """Route the host API through the CPU oracle -- TEST ONLY (the product has no CPU path)."""
    from oracle.adapter import OracleBackend
    from openpcseg_amd import native
    be = OracleBackend()
    monkeypatch.setattr(native, "_BACKEND", be)
    return be


@pytest.fixture(scope="session")
def ref_backend():
    """The reference's own compiled CPU backend (oracle/_ref); skipped where it is not built."""
    from oracle import build_ref
    if build_ref.load() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from oracle.adapter import RefBackend
    return RefBackend()


@pytest.fixture(scope="session")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from openpcseg_amd import native
    return native.HipBackend()
