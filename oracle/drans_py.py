"""ORACLE (test infrastructure only) -- restatement of the opt-in device wire format "DGR1"
(include/dcvc_hip_rans.h) on top of the oracle's single-stream rANS (oracle/rans_py.py, or the C
restatement oracle/rans_ref.c through `encode_fn`).  The format is this repo's own (SURVEY 8f-3:
"changes the wire format -> must be opt-in"), so there is no reference to pin it against: the
test is that the GPU kernels produce exactly these bytes and that both directions round-trip.
A lane whose stream carries an escape of more than 8 nibbles is corrupt: rans_py.Decoder raises
ValueError there, the device decoder sets DCVC_DRANS_BAD_STREAM, the host decoder returns
DCVC_RANS_E_STREAM -- the same limit in all four decoders.
"""
import struct

import numpy as np

from . import rans_py

MAGIC = b"DGR1"


def default_lanes(n):
    l = (n + 511) // 512
    l = (l + 63) // 64 * 64
    return max(64, min(1024, l))


def encode_section(symbols, indexes, cdfs, sizes, offsets, lanes=None, encode_fn=None):
    """One plane -> section bytes.  Lane j codes symbols j, j+L, ... as an independent stream."""
    symbols, indexes = np.asarray(symbols), np.asarray(indexes)
    n = symbols.size
    L = lanes or default_lanes(n)
    enc = encode_fn or (lambda s, i: rans_py.encode([(s, i, cdfs, sizes, offsets)]))
    streams = [enc(symbols[j::L], indexes[j::L]) for j in range(L)]
    words = [len(s) // 4 for s in streams]
    return struct.pack("<2I", n, L) + struct.pack("<%dI" % L, *words) + b"".join(streams)


def encode_picture(planes, **kw):
    """planes: list of (symbols, indexes, cdfs, sizes, offsets) in coding order."""
    return MAGIC + b"".join(encode_section(*p, **kw) for p in planes)


def decode_section(data, pos, indexes, cdfs, sizes, offsets):
    """-> (symbols, new position in bytes)."""
    n, L = struct.unpack_from("<2I", data, pos)
    words = struct.unpack_from("<%dI" % L, data, pos + 8)
    indexes = np.asarray(indexes)
    assert indexes.size == n
    out = np.zeros(n, np.int64)
    p = pos + 8 + 4 * L
    for j in range(L):
        seg = data[p : p + 4 * words[j]]
        p += 4 * words[j]
        if indexes[j::L].size:
            out[j::L] = rans_py.Decoder(seg).decode(indexes[j::L], cdfs, sizes, offsets)
    return out, p
