"""Golden byte strings for the rANS wire format, produced by the independent pure-Python
restatement (oracle/rans_py.py).  The reference cannot produce them here (its coder needs
the absent ryg_rans header), so these pin the two C implementations to the restated format,
not to the reference's binary: "parity unpinned" at this boundary."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rans_py  # noqa: E402


def main():
    t = np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))
    cdf, ln, off = t["dmc_scale_cdf"], t["dmc_scale_len"], t["dmc_scale_off"]
    g = np.random.default_rng(11)
    fx = {}
    cases = {
        "small": (g.integers(-3, 4, 200), g.integers(0, 256, 200)),
        "bypass": (g.integers(-300, 300, 300), g.integers(0, 40, 300)),
        "huge": (np.array([0, 70000, -70000, 1, 2 ** 20, -(2 ** 20), 5]), np.array([0, 1, 2, 255, 128, 7, 9])),
        "empty": (np.zeros(0, np.int64), np.zeros(0, np.int64)),
        "one": (np.array([0]), np.array([100])),
    }
    for name, (sym, idx) in cases.items():
        data = rans_py.encode([(sym.tolist(), idx.tolist(), cdf, ln, off)])
        assert rans_py.Decoder(data).decode(idx.tolist(), cdf, ln, off) == sym.tolist()
        fx[name + "_sym"], fx[name + "_idx"] = sym.astype(np.int32), idx.astype(np.int32)
        fx[name + "_bytes"] = np.frombuffer(data, np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rans_bytes.npz"), **fx)
    print({k: v.shape for k, v in fx.items() if k.endswith("bytes")})


if __name__ == "__main__":
    main()
