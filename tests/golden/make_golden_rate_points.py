"""Fixture for the rate-point selection (video_coder.py:181-197): the reference's own interpolate_log
(DCVC_HEM/src/utils/common.py:23-31) evaluated in THIS container on a few anchor sets.  -> tests/golden/rate_points.npz
    python tests/golden/make_golden_rate_points.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import refimport

refimport.load()
from DCVC_HEM.src.utils.common import interpolate_log  # noqa: E402

cases = [(0.5, 1.8, 6), (0.3, 2.7, 4), (1.0, 1.0001, 2), (0.05, 12.0, 9), (0.9, 1.1, 64)]
fx = {"lo": np.array([c[0] for c in cases]), "hi": np.array([c[1] for c in cases]), "num": np.array([c[2] for c in cases])}
for k, (lo, hi, n) in enumerate(cases):
    fx[f"dec_{k}"] = interpolate_log(lo, hi, n)
    fx[f"asc_{k}"] = interpolate_log(lo, hi, n, decending=False)
np.savez_compressed(os.path.join(HERE, "rate_points.npz"), **fx)
print("wrote rate_points.npz")
