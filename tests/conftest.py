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


def _cache_synthetic_frames():
    """vcm_ts_amd.synthetic.frames takes 5 s for eight 1080p pictures and a dozen bench-size tests ask for the same
    ones: keep the longest sequence generated per (seed, size, noise) and hand out copies of its prefix (frame t does
    not depend on how many frames follow it: the generator draws them in order)."""
    import vcm_ts_amd.synthetic as S

    plain, kept = S.frames, {}

    def frames(seed, n_frames, h, w, noise=1.0 / 255.0):
        key = (int(seed), int(h), int(w), float(noise))
        have = kept.get(key)
        if have is None or have.shape[0] < n_frames:
            have = kept[key] = plain(seed, n_frames, h, w, noise)
        return have[:n_frames].copy()

    frames.__doc__ = plain.__doc__
    S.frames = frames


_cache_synthetic_frames()
