"""ctypes bindings of the two C-ABI libraries (include/dcvc_hip.h, include/dcvc_rans.h).

There is no fallback: if a library is missing the import of the product path fails loudly
(`LibraryMissing`) -- build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C vcm_ts_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


class LibraryMissing(RuntimeError):
    pass


class KernelError(RuntimeError):
    pass


def _load(name):
    # DCVC_HIP_LIB: developer override used by the kernel probes (tools/conv_probe.py) only
    path = os.environ.get("DCVC_HIP_LIB") if name == "libdcvc_hip.so" and os.environ.get("DCVC_HIP_LIB") else \
        os.path.join(CSRC, name)
    if not os.path.exists(path):
        raise LibraryMissing(f"{path} not built; run `make -C {CSRC}` (no CPU fallback exists)")
    return C.CDLL(path)


# ----------------------------------------------------------------------------- rans (host)
_rans = None


def rans():
    global _rans
    if _rans is None:
        L = _load("libdcvc_rans.so")
        L.dcvc_rans_encoder_create.restype = C.c_void_p
        L.dcvc_rans_encoder_destroy.argtypes = [C.c_void_p]
        L.dcvc_rans_encoder_reset.argtypes = [C.c_void_p]
        L.dcvc_rans_encoder_encode_with_indexes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                                            C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.dcvc_rans_encoder_flush_bound.argtypes = [C.c_void_p]
        L.dcvc_rans_encoder_flush_bound.restype = C.c_int64
        L.dcvc_rans_encoder_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.dcvc_rans_encoder_flush.restype = C.c_int64
        L.dcvc_rans_decoder_create.restype = C.c_void_p
        L.dcvc_rans_decoder_destroy.argtypes = [C.c_void_p]
        L.dcvc_rans_decoder_set_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.dcvc_rans_decoder_decode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dcvc_pmf_to_quantized_cdf.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        _rans = L
    return _rans


# ----------------------------------------------------------------------------- hip kernels
class Seg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("cs", C.c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [
        ("seg", Seg * 3), ("nseg", C.c_int32), ("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("in_act", C.c_int32), ("in_slope", C.c_float), ("wpack", C.c_void_p), ("bpack", C.c_void_p),
        ("ks", C.c_int32), ("stride", C.c_int32), ("Cout", C.c_int32), ("Cout_pad", C.c_int32),
        ("out", C.c_void_p), ("out_cs", C.c_int32), ("out_act", C.c_int32), ("out_slope", C.c_float),
        ("pixel_shuffle", C.c_int32), ("res", C.c_void_p), ("res_cs", C.c_int32), ("res_gate", C.c_void_p),
        ("res2", C.c_void_p), ("res2_cs", C.c_int32), ("precision", C.c_int32), ("status", C.c_void_p),
        ("chan_partial", C.c_void_p), ("tile_row0", C.c_int32), ("tile_rows", C.c_int32), ("pair_taps", C.c_int32),
    ]


class DualPriorArgs(C.Structure):
    _fields_ = [
        ("y", C.c_void_p), ("y_cs", C.c_int32), ("fusion", C.c_void_p), ("fusion_cs", C.c_int32),
        ("spatial", C.c_void_p), ("spatial_cs", C.c_int32), ("params", C.c_void_p), ("params_cs", C.c_int32),
        ("y_hat", C.c_void_p), ("y_q", C.c_void_p), ("y_res", C.c_void_p), ("scales_hat", C.c_void_p),
        ("sym", C.c_void_p), ("idx", C.c_void_p), ("out", C.c_void_p), ("out_cs", C.c_int32),
        ("q_basic", C.c_void_p), ("q_scale", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("C", C.c_int32), ("step", C.c_int32), ("idx_edges", C.c_void_p), ("forced_q", C.c_void_p),
    ]


class ConvBwdArgs(C.Structure):
    _fields_ = [
        ("dout", C.c_void_p), ("dout_cs", C.c_int32), ("out", C.c_void_p), ("out_cs", C.c_int32),
        ("res", C.c_void_p), ("res_cs", C.c_int32), ("gate", C.c_void_p), ("res2", C.c_void_p), ("res2_cs", C.c_int32),
        ("dres", C.c_void_p), ("dres_cs", C.c_int32), ("dres2", C.c_void_p), ("dres2_cs", C.c_int32),
        ("dpre", C.c_void_p), ("dpre_cs", C.c_int32), ("zs", C.c_int32), ("Hd", C.c_int32), ("Wd", C.c_int32),
        ("N", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("pixel_shuffle", C.c_int32),
        ("act", C.c_int32), ("slope", C.c_float),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_cs", C.c_int32), ("C", C.c_int32), ("in_act", C.c_int32), ("in_slope", C.c_float),
        ("dpre", C.c_void_p), ("dpre_cs", C.c_int32), ("zs", C.c_int32), ("Hd", C.c_int32), ("Wd", C.c_int32),
        ("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("Cout", C.c_int32), ("ks", C.c_int32), ("stride", C.c_int32), ("dw", C.c_void_p), ("Cin_total", C.c_int32),
        ("cin_offset", C.c_int32), ("scratch", C.c_void_p), ("scratch_floats", C.c_int64), ("overwrite", C.c_int32),
        ("db", C.c_void_p), ("precision", C.c_int32),
    ]


class PackJob(C.Structure):  # dcvc_pack_job (include/dcvc_hip_grad.h)
    _fields_ = [
        ("w", C.c_void_p), ("b", C.c_void_p), ("Cout", C.c_int32), ("Cin_total", C.c_int32), ("ks", C.c_int32),
        ("nseg", C.c_int32), ("seg_C", C.c_int32 * 3), ("cin_offset", C.c_int32), ("pixel_shuffle", C.c_int32),
        ("precision", C.c_int32), ("transposed", C.c_int32), ("wpack", C.c_void_p), ("bpack", C.c_void_p),
    ]


class DualPriorBwdArgs(C.Structure):
    _fields_ = [
        ("y", C.c_void_p), ("y_cs", C.c_int32), ("fusion", C.c_void_p), ("fusion_cs", C.c_int32),
        ("y_hat", C.c_void_p), ("dout", C.c_void_p), ("dout_cs", C.c_int32), ("dy_res", C.c_void_p),
        ("dscales_hat", C.c_void_p), ("dparams", C.c_void_p), ("dparams_cs", C.c_int32),
        ("dspatial", C.c_void_p), ("dspatial_cs", C.c_int32), ("dy", C.c_void_p), ("dy_cs", C.c_int32),
        ("dfusion", C.c_void_p), ("dfusion_cs", C.c_int32), ("dq_plane", C.c_void_p),
        ("q_basic", C.c_void_p), ("q_scale", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("C", C.c_int32), ("step", C.c_int32),
    ]


PRECISIONS = {"fp32": 0, "fp16x3": 1}

_hip = None
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

_SIGS = {
    "dcvc_conv2d": [C.POINTER(ConvArgs), vp],
    "dcvc_conv_pack_weights": [vp, vp, i32, i32, i32, vp, i32, i32, vp, vp],
    "dcvc_conv_pack_weights_paired": [vp, vp, i32, i32, vp, vp],
    "dcvc_conv2d_small": [C.POINTER(ConvArgs), vp],
    "dcvc_conv_small_pack_weights": [vp, vp, i32, i32, i32, vp, vp, vp],
    "dcvc_conv2d_k32": [C.POINTER(ConvArgs), vp],
    "dcvc_conv_k32_pack_weights": [vp, vp, i32, i32, i32, vp, i32, vp, vp],
    "dcvc_conv_k32_set_waves": [i32],
    "dcvc_warp": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "dcvc_up2": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, f32, vp],
    "dcvc_down2": [vp, i32, vp, i32, i32, i32, i32, i32, f32, i32, vp],
    "dcvc_maxpool2": [vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "dcvc_copy_channels": [vp, i32, vp, i32, i64, i32, vp],
    "dcvc_nchw_to_nhwc": [vp, vp, i32, i32, i32, i32, i32, vp],
    "dcvc_nhwc_to_nchw": [vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "dcvc_channel_mean": [vp, i32, vp, vp, i32, i32, i32, vp],
    "dcvc_se_gate": [vp, vp, vp, vp, i32, i32, i32, vp],
    "dcvc_channel_mean_finish": [vp, i32, i32, vp, i32, i32, i32, vp],
    "dcvc_scale_channels": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "dcvc_round_symbols": [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp],
    "dcvc_symbols_to_nhwc": [vp, vp, i32, i32, i32, i32, i32, vp],
    "dcvc_scale_indexes": [vp, vp, i64, vp, vp],
    "dcvc_dual_prior_enc": [C.POINTER(DualPriorArgs), vp],
    "dcvc_dual_prior_dec_index": [C.POINTER(DualPriorArgs), vp],
    "dcvc_dual_prior_dec_apply": [C.POINTER(DualPriorArgs), vp],
    "dcvc_build_scale_cdfs": [vp, i32, i32, vp, vp, vp, vp],
    "dcvc_build_factorized_cdfs": [vp, i32, vp, vp, vp, vp],
    "dcvc_scale_bits": [vp, vp, vp, vp, i32, i32, i64, vp],
    "dcvc_factorized_bits": [vp, i32, vp, vp, vp, i32, i32, i32, vp],
    "dcvc_sq_err": [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp],
    # include/dcvc_hip_grad.h
    "dcvc_conv_pack_weights_dev": [vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, vp, vp, vp],
    "dcvc_pack_plan_create": [C.POINTER(PackJob), i32, C.POINTER(C.c_void_p)],
    "dcvc_pack_plan_run": [vp, vp],
    "dcvc_pack_plan_destroy": [vp],
    "dcvc_conv_bwd_prologue": [C.POINTER(ConvBwdArgs), vp],
    "dcvc_conv_wgrad": [C.POINTER(WgradArgs), vp],
    "dcvc_channel_dot": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "dcvc_mask_accumulate": [vp, i32, vp, i32, f32, vp, i32, i64, i32, vp],
    "dcvc_add_planes": [vp, i32, vp, i32, vp, i32, i64, i32, vp],
    "dcvc_warp_bwd": [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp],
    "dcvc_up2_bwd": [vp, i32, vp, i32, i32, i32, i32, i32, f32, vp],
    "dcvc_down2_bwd": [vp, i32, vp, i32, i32, i32, i32, i32, f32, vp],
    "dcvc_maxpool2_bwd": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "dcvc_se_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "dcvc_add_channel_vec": [vp, i32, vp, f32, i32, i32, i32, vp],
    "dcvc_scale_channels_bwd": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "dcvc_q_finish": [vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "dcvc_dual_prior_bwd": [C.POINTER(DualPriorBwdArgs), vp],
    "dcvc_scale_bits_bwd": [vp, vp, vp, vp, vp, i32, i32, i64, vp],
    "dcvc_factorized_bits_bwd": [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, vp],
    "dcvc_sq_err_bwd": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp],
    # include/dcvc_hip_rans.h
    "dcvc_drans_encode": [vp, vp, i32, i32, i64, vp, i32, i32, vp, vp, i32, vp, i64, vp, i64, vp, vp, vp, vp],
    "dcvc_drans_decode": [vp, i64, vp, vp, vp, i32, i32, i64, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp],
    "dcvc_drans_build_lut": [vp, i32, i32, vp, vp],
}

HIP_SYMBOLS = sorted(list(_SIGS) + ["dcvc_cdf_table_cols", "dcvc_conv_pack_size", "dcvc_conv_pack_size_paired", "dcvc_conv_small_pack_bytes", "dcvc_conv_k32_pack_bytes", "dcvc_conv_tile_rows", "dcvc_conv_chan_partial_parts", "dcvc_hip_version", "dcvc_conv_wgrad_scratch_min",
                                    "dcvc_drans_default_lanes", "dcvc_drans_scratch_words"])
RANS_SYMBOLS = [
    "dcvc_rans_encoder_create", "dcvc_rans_encoder_destroy", "dcvc_rans_encoder_reset",
    "dcvc_rans_encoder_encode_with_indexes", "dcvc_rans_encoder_flush_bound", "dcvc_rans_encoder_flush",
    "dcvc_rans_decoder_create", "dcvc_rans_decoder_destroy", "dcvc_rans_decoder_set_stream",
    "dcvc_rans_decoder_decode_stream", "dcvc_pmf_to_quantized_cdf",
]


def hip():
    global _hip
    if _hip is None:
        L = _load("libdcvc_hip.so")
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = C.c_int
        L.dcvc_pack_plan_destroy.restype = None
        L.dcvc_conv_pack_size.argtypes = [i32, i32, i32, vp, C.POINTER(i32)]
        L.dcvc_conv_pack_size.restype = i64
        L.dcvc_conv_pack_size_paired.argtypes = [i32, i32, C.POINTER(i32)]
        L.dcvc_conv_pack_size_paired.restype = i64
        L.dcvc_cdf_table_cols.argtypes = []
        L.dcvc_cdf_table_cols.restype = i32
        L.dcvc_conv_chan_partial_parts.argtypes = [i32, i32, i32, i32]
        L.dcvc_conv_chan_partial_parts.restype = i32
        L.dcvc_conv_small_pack_bytes.argtypes = [i32, i32, i32, vp]
        L.dcvc_conv_small_pack_bytes.restype = i64
        L.dcvc_conv_tile_rows.argtypes = [i32, i32]
        L.dcvc_conv_tile_rows.restype = i32
        L.dcvc_conv_k32_pack_bytes.argtypes = [i32, i32, i32, vp, C.POINTER(i32)]
        L.dcvc_conv_k32_pack_bytes.restype = i64
        L.dcvc_hip_version.restype = C.c_char_p
        L.dcvc_conv_wgrad_scratch_min.argtypes = [i32, i32, i32]
        L.dcvc_conv_wgrad_scratch_min.restype = i64
        L.dcvc_drans_default_lanes.argtypes = [i64]
        L.dcvc_drans_default_lanes.restype = i32
        L.dcvc_drans_scratch_words.argtypes = [i64, i32]
        L.dcvc_drans_scratch_words.restype = i64
        _hip = L
    return _hip


def check(code, what):
    if code != 0:
        raise KernelError(f"{what} failed with status {code}")
