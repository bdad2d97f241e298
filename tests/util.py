"""Shared helpers for the test-suite (the only place, with bench.py's cpu_baseline leg and
__graft_entry__.smoke(), that may touch oracle/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from vcm_ts_amd.params import dmc_spec, intra_spec, seeded_state_dict  # noqa: E402

_W = {}


def oracle_weights(kind, seed=0, gain=None):
    """Name-seeded weights; (seed 5, gain 1.2) is the second weight set of tests/golden/seq_128x192_w5.npz."""
    key = (kind, seed, gain)
    if key not in _W:
        kw = {"seed": seed} if gain is None else {"seed": seed, "gain": gain}
        _W[key] = seeded_state_dict(dmc_spec() if kind == "dmc" else intra_spec(), **kw)
    return _W[key]


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def stats(t):
    # float64 reductions where the tensor lives (a 64-channel 1088x1920 DPB feature is a gigabyte in double: on the host
    # these four reductions were two thirds of the bench-size tests' time); the summation order inside float64 is
    # far below every tolerance the results are compared with
    t = t.detach().double()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item(), t.sum().item()], np.float64)


def crop(t):
    return t.detach()[..., :8, :8].contiguous().cpu().numpy().astype(np.float32)
