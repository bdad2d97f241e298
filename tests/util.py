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


def oracle_weights(kind):
    if kind not in _W:
        _W[kind] = seeded_state_dict(dmc_spec() if kind == "dmc" else intra_spec())
    return _W[kind]


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def stats(t):
    t = t.detach().double().cpu()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item(), t.sum().item()], np.float64)


def crop(t):
    return t.detach()[..., :8, :8].contiguous().cpu().numpy().astype(np.float32)
