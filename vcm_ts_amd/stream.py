"""Geometry helpers and the `.bin` wire format of one coded picture.

Same functions, argument order and bytes as the reference's
/root/reference/DCVC_HEM/src/utils/stream_helper.py (:24-46 geometry and q rounding,
:103-144 encode_i/decode_i/encode_p/decode_p).  All header fields are big-endian:

    I picture:  >II height, width   >H q_index   >I payload length   payload
    P picture:  >HH mv_y_q_index, y_q_index      >I payload length   payload
"""
from __future__ import annotations

import os
import struct

import numpy as np
import torch

_I_HEAD = struct.Struct(">IIHI")
_P_HEAD = struct.Struct(">HHI")


def _ceil_to(v, p):
    return -(-v // p) * p


def get_padding_size(height, width, p=64):
    """(left, right, top, bottom): the picture is padded on the right and bottom only."""
    return 0, _ceil_to(width, p) - width, 0, _ceil_to(height, p) - height


def get_downsampled_shape(height, width, p):
    return _ceil_to(height, p) // p, _ceil_to(width, p) // p


def get_rounded_q(q_scale):
    """Clip to [0.01, 655] and snap to 1/100 steps: returns (q_scale, uint16 index)."""
    q_index = int(np.round(np.clip(q_scale, 0.01, 655.0) * 100))
    return q_index / 100, q_index


def get_state_dict(ckpt_path):
    ckpt = torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=True)
    for key in ("state_dict", "net"):
        if key in ckpt:
            ckpt = ckpt[key]
    return {(k[7:] if k.startswith("module.") else k): v for k, v in ckpt.items()}


def filesize(filepath) -> int:
    if not os.path.isfile(filepath):
        raise ValueError(f'Invalid file "{filepath}".')
    return os.stat(filepath).st_size


def encode_i(height, width, q_index, bit_stream, output):
    with open(output, "wb") as f:
        f.write(_I_HEAD.pack(height, width, q_index, len(bit_stream)))
        f.write(bit_stream)


def _payload(f, n, inputpath):
    """The n payload bytes a header announces.  The reference returns whatever is left (f.read(n)) and lets the entropy
    decoder run off the end; a file shorter than its own header says is refused here, by name."""
    data = f.read(n)
    if len(data) != n:
        raise ValueError(f'"{inputpath}" is truncated: its header announces {n} payload bytes, {len(data)} are there')
    return data


def _header(f, head, inputpath):
    raw = f.read(head.size)
    if len(raw) != head.size:
        raise ValueError(f'"{inputpath}" is truncated: {len(raw)} of {head.size} header bytes')
    return head.unpack(raw)


def decode_i(inputpath):
    with open(inputpath, "rb") as f:
        height, width, q_index, n = _header(f, _I_HEAD, inputpath)
        return height, width, q_index, _payload(f, n, inputpath)


def encode_p(string, mv_y_q_index, y_q_index, output):
    with open(output, "wb") as f:
        f.write(_P_HEAD.pack(mv_y_q_index, y_q_index, len(string)))
        f.write(string)


def decode_p(inputpath):
    with open(inputpath, "rb") as f:
        mv_y_q_index, y_q_index, n = _header(f, _P_HEAD, inputpath)
        return mv_y_q_index, y_q_index, _payload(f, n, inputpath)
