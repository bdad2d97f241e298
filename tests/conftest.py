import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared libraries are build products (git-ignored): build them once if a fresh checkout
    has none (same commands as __graft_entry__.build())."""
    import subprocess

    csrc = os.path.join(ROOT, "vcm_ts_amd", "csrc")
    if not all(os.path.exists(os.path.join(csrc, n)) for n in ("libdcvc_hip.so", "libdcvc_rans.so")):
        subprocess.check_call(["make", "-C", csrc, "-j4"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_ref.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
