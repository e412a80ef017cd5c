import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Built artefacts are git-ignored; if a fresh checkout lacks them and the toolchain is here, build them
    once (the GPU box always receives them prebuilt with the tree)."""
    need = [os.path.join(ROOT, "iterative_solvers_amd", "libits_hip.so"),
            os.path.join(ROOT, "iterative_solvers_amd", "host", "its_solve"),
            os.path.join(ROOT, "oracle", "libits_oracle.so")]
    if all(os.path.exists(p) for p in need) or not os.path.exists("/opt/rocm/bin/hipcc"):
        return
    import subprocess
    for d in ("iterative_solvers_amd/csrc", "iterative_solvers_amd/host"):
        subprocess.run(["make", "-C", os.path.join(ROOT, d)], check=False, capture_output=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=False, capture_output=True)


@pytest.fixture(scope="session")
def gpu_handle():
    """One library handle on cuda:0 for the kernel-level GPU tests."""
    from iterative_solvers_amd import capi
    h = capi.Handle(0)
    yield h
    h.close()
