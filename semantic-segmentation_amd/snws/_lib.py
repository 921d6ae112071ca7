"""ctypes binding of libsnws.so (the C ABI declared in include/snws.h).

The library is built in-tree (semantic-segmentation_amd/lib/libsnws.so) by __graft_entry__.build() or
`make -C semantic-segmentation_amd/csrc`.  There is NO fallback: if the library is missing, or a call
returns a non-zero status, an exception is raised.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p, POINTER, Structure

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNWS_LIB = <file name inside lib/>: load another in-tree build of the library (same-box A/B runs of two kernel versions)
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", os.path.basename(os.environ.get("SNWS_LIB", "libsnws.so")))

SNWS_F32, SNWS_BF16, SNWS_F64, SNWS_I64, SNWS_I32, SNWS_U8 = 0, 1, 2, 3, 4, 5
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP = -1, -2, -3


class ConvDesc(Structure):
    _fields_ = [("N", c_int), ("H", c_int), ("W", c_int), ("C_in", c_int), ("C_out", c_int), ("kh", c_int),
                ("kw", c_int), ("stride", c_int), ("pad", c_int), ("dil", c_int), ("groups", c_int),
                ("ldx", c_int64), ("ldy", c_int64)]


class SnBwdSeg(Structure):
    _fields_ = [("c_begin", c_int), ("c_count", c_int), ("x", c_void_p), ("ldx", c_int64), ("ctx", c_void_p),
                ("beta", c_void_p), ("part", c_void_p), ("relu", c_int)]


class SnFwdItem(Structure):
    """snws_sn_fwd_item: one SwitchNorm call of a batched forward launch."""
    _fields_ = [("part", c_void_p), ("shift", c_void_p), ("ctx", c_void_p), ("pack", c_void_p), ("gamma", c_void_p),
                ("beta", c_void_p), ("mean_weight", c_void_p), ("var_weight", c_void_p), ("running_mean", c_void_p),
                ("running_var", c_void_p), ("x", c_void_p), ("y", c_void_p), ("ldx", c_int64), ("ldy", c_int64),
                ("T", c_int), ("shift_per_image", c_int)]


class SnBwdItem(Structure):
    """snws_sn_bwd_item: one SwitchNorm call of a batched backward launch."""
    _fields_ = [("part", c_void_p), ("ctx", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("mean_weight", c_void_p), ("var_weight", c_void_p), ("ctxb", c_void_p), ("packb", c_void_p),
                ("dgamma", c_void_p), ("dbeta", c_void_p), ("dmean_weight", c_void_p), ("dvar_weight", c_void_p),
                ("dx_sum", c_void_p), ("dy", c_void_p), ("x", c_void_p), ("dx", c_void_p), ("lddy", c_int64),
                ("ldx", c_int64), ("lddx", c_int64), ("T", c_int), ("accumulate_a", c_int), ("accumulate_b", c_int),
                ("reserved", c_int)]


SN_BATCH_MAX = 8
P = c_void_p
_SIGNATURES = {
    "snws_version": (c_int, []),
    "snws_last_error": (c_char_p, []),
    "snws_confusion_hist": (c_int, [P, c_int, P, c_int, c_int64, c_int, P, P]),
    "snws_confusion_hist_argmax": (c_int, [P, c_int, P, c_int, c_int64, c_int64, c_int, P, P]),
    "snws_ws_fwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    "snws_conv_pack": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P]),
    "snws_ws_bwd": (c_int, [P, P, c_int64, P, c_int, c_int, c_int, c_int, c_int, P]),
    "snws_conv_pack_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "snws_conv_out_size": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "snws_conv_direct_fwd": (c_int, [POINTER(ConvDesc), c_int, P, P, P, P, P]),
    "snws_conv_direct_dgrad": (c_int, [POINTER(ConvDesc), c_int, P, P, P, P]),
    "snws_conv_direct_wgrad": (c_int, [POINTER(ConvDesc), c_int, P, P, P, P, P]),
    "snws_conv_mfma_tiles": (c_int, [c_int, c_int]),
    "snws_conv_mfma_fwd": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    "snws_conv_mfma_dgrad": (c_int, [POINTER(ConvDesc), P, P, P, P]),
    "snws_conv_mfma_dgrad_sn": (c_int, [POINTER(ConvDesc), P, P, P, POINTER(SnBwdSeg), c_int, P]),
    "snws_conv_mfma_wgrad": (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    "snws_conv_mfma_wgrad_workspace_bytes": (c_int64, [POINTER(ConvDesc)]),
    "snws_conv_mfma_wgrad_det": (c_int, [POINTER(ConvDesc), P, P, P, P, P, c_int64, P]),
    "snws_conv_f32_tiles": (c_int, [c_int, c_int]),
    "snws_conv_f32_fwd": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    "snws_conv_f32_flip": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "snws_conv_f32_dgrad": (c_int, [POINTER(ConvDesc), P, P, P, P]),
    "snws_conv_f32_wgrad_workspace_bytes": (c_int64, [POINTER(ConvDesc)]),
    "snws_conv_f32_wgrad": (c_int, [POINTER(ConvDesc), P, P, P, P, P, c_int64, P]),
    "snws_sn_ctx_floats": (c_int64, [c_int, c_int]),
    "snws_sn_ctxb_floats": (c_int64, [c_int, c_int]),
    "snws_sn_stats_tiles": (c_int, [c_int, c_int64, c_int]),
    "snws_sn_stats": (c_int, [P, c_int, c_int, c_int64, c_int, c_int64, P, P, P]),
    "snws_sn_finalize": (c_int, [P, c_int, P, c_int, c_int, c_int64, c_int, P, P]),
    "snws_sn_mix_a": (c_int, [P, P, c_int, c_int, P]),
    "snws_sn_mix_b": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_float, c_float, c_int, c_int, c_int, P]),
    "snws_sn_apply": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int, P, P, c_int, c_int64, c_int, c_int, P]),
    "snws_sn_fwd_a_batched": (c_int, [POINTER(SnFwdItem), c_int, c_int, c_int64, c_int, P]),
    "snws_sn_mix_b_batched": (c_int, [POINTER(SnFwdItem), c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_int, P]),
    "snws_sn_apply_batched": (c_int, [POINTER(SnFwdItem), c_int, c_int, c_int, c_int64, c_int, c_int, P]),
    "snws_sn_bwd_a_batched": (c_int, [POINTER(SnBwdItem), c_int, c_int, c_int, P]),
    "snws_sn_bwd_mix_b_batched": (c_int, [POINTER(SnBwdItem), c_int, c_int, c_int64, c_int, c_float, c_int, c_int, P]),
    "snws_sn_bwd_apply_batched": (c_int, [POINTER(SnBwdItem), c_int, c_int, c_int, c_int64, c_int, c_int, P]),
    "snws_sn_noalloc_max_floats": (c_int64, []),
    "snws_sn_noalloc_floats": (c_int64, [c_int, c_int64, c_int]),
    "snws_sn_apply_noalloc": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int, P, P, P, P, P, P, c_int, c_int64, c_int,
                                      c_float, c_float, c_int, c_int, c_int, c_int, P]),
    "snws_conv_mfma_dgrad_acc": (c_int, [POINTER(ConvDesc), P, P, P, P]),
    "snws_add_inplace": (c_int, [P, c_int64, P, c_int64, c_int, c_int64, c_int, P]),
    "snws_gemm1x1_supported": (c_int, [c_int, c_int]),
    "snws_gemm1x1": (c_int, [P, c_int64, P, P, P, c_int64, c_int64, c_int, c_int, P]),
    "snws_gemm1x1_acc": (c_int, [P, c_int64, P, P, c_int64, c_int64, c_int, c_int, P]),
    "snws_gemm1x1_stat_tiles": (c_int, [c_int64]),
    "snws_gemm1x1_stats": (c_int, [P, c_int64, P, P, P, c_int64, c_int64, c_int, c_int, c_int, c_int64, P, P]),
    "snws_ocr_tiles": (c_int, [c_int64, c_int]),
    "snws_ocr_gather_stats": (c_int, [P, c_int64, P, c_int, c_int64, c_int, c_float, P]),
    "snws_ocr_gather_fwd": (c_int, [P, c_int64, P, c_int64, P, P, c_int, c_int64, c_int, c_int, c_float, P]),
    "snws_ocr_gather_bwd": (c_int, [P, c_int64, P, c_int64, P, P, P, P, c_int64, P, c_int, c_int64, c_int, c_int, c_float, P]),
    "snws_ocr_gather_bwd_acc": (c_int, [P, c_int64, P, c_int64, P, P, P, P, c_int64, P, c_int, c_int64, c_int, c_int, c_float, P]),
    "snws_ocr_attend_fwd": (c_int, [P, c_int64, P, P, P, c_int64, c_int, c_int64, c_int, c_int, c_float, P]),
    "snws_ocr_attend_bwd": (c_int, [P, c_int64, P, c_int64, P, P, P, c_int64, P, P, c_int, c_int64, c_int, c_int, c_float, P]),
    "snws_ocr32_runs": (c_int, [c_int64]),
    "snws_ocr32_gather_fwd": (c_int, [P, c_int64, P, c_int64, P, P, c_int, c_int64, c_int, c_int, c_float, P]),
    "snws_ocr32_gather_bwd": (c_int, [P, c_int64, P, c_int64, P, P, P, P, c_int64, P, c_int64, c_int, c_int64, c_int, c_int,
                                      c_float, c_int, P]),
    "snws_ocr32_attend_fwd": (c_int, [P, c_int64, P, P, P, c_int64, c_int, c_int64, c_int, c_int, c_float, P]),
    "snws_ocr32_attend_bwd": (c_int, [P, c_int64, P, c_int64, P, P, P, c_int64, P, P, c_int, c_int64, c_int, c_int, c_float, P]),
    "snws_sn_head_tiles": (c_int, [c_int64, c_int, c_int]),
    "snws_sn_apply_head": (c_int, [P, c_int64, P, P, P, P, P, c_int, c_int64, c_int, c_int, P]),
    "snws_sn_head_bwd_reduce": (c_int, [P, P, c_int64, P, P, P, P, P, P, c_int, c_int64, c_int, c_int, P]),
    "snws_sn_head_bwd_apply": (c_int, [P, P, c_int64, P, P, P, P, P, c_int64, c_int, c_int64, c_int, c_int, P]),
    "snws_partial_sum": (c_int, [P, c_int, c_int, P, c_int, P]),
    "snws_partial_sum_batched": (c_int, [P, c_int, c_int, c_int, P, c_int, P]),
    "snws_sn_bwd_reduce": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int, P, P, P, c_int, c_int64, c_int, c_int,
                                   P, P]),
    "snws_sn_bwd_mix_a": (c_int, [P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "snws_sn_bwd_mix_b": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int64, c_int, c_float, c_int, c_int, c_int, P]),
    "snws_sn_bwd_apply": (c_int, [P, c_int64, P, c_int64, P, c_int64, P, c_int64, c_int, P, P, P, c_int, c_int64,
                                  c_int, c_int, P]),
    "snws_sn_bwd_apply_res": (c_int, [P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int64, c_int, P, P, P, c_int,
                                      c_int64, c_int, P]),
    "snws_upconv_fwd": (c_int, [P, c_int64, P, c_int64, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "snws_upconv_stat_tiles": (c_int, [c_int, c_int]),
    "snws_upconv_fwd_mode": (c_int, [c_int]),
    "snws_upconv_bwd_mode": (c_int, [c_int]),
    "snws_upconv_fwd_stats": (c_int, [P, c_int64, P, c_int64, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "snws_upconv_bwd": (c_int, [P, c_int64, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "snws_upconv_bwd_acc": (c_int, [P, c_int64, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "snws_resample_fwd": (c_int, [P, c_int64, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "snws_resample_bwd": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, P]),
    "snws_bn_pack": (c_int, [P, c_int, c_int64, c_int, P, P]),
    "snws_bn_coeff": (c_int, [P, P, P, P, P, c_int, c_int64, c_int, c_float, c_float, c_int, c_int, P]),
    "snws_bn_bwd_pack": (c_int, [P, c_int, c_int, c_int, P, P, P, P]),
    "snws_bn_bwd_pack_post": (c_int, [P, c_int, c_int, c_int, P, P, P, P, P]),
    "snws_sn_apply_post": (c_int, [P, c_int64, P, c_int64, c_int, P, P, P, c_int, c_int64, c_int, c_int, P]),
    "snws_sn_bwd_apply_post": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int, P, P, P, P, c_int, c_int64, c_int, c_int, P]),
    "snws_bn_bwd_coeff": (c_int, [P, P, P, c_int, c_int, c_float, c_int, P]),
    "snws_tile_minmax_blocks": (c_int, [c_int64]),
    "snws_tile_minmax_scale": (c_int, [P, P, c_int, c_int, c_int64, P, P, P]),
    "snws_label_remap": (c_int, [P, c_int, P, c_int64, P]),
    "snws_nchw_to_nhwc_pad": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "snws_adamw_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64, c_float, P]),
    "snws_comm_unique_id": (c_int, [P]),
    "snws_comm_init": (c_int, [P, c_int, c_int, P]),
    "snws_comm_allreduce_f32": (c_int, [P, P, c_int64, P]),
    "snws_comm_allreduce_i64": (c_int, [P, P, c_int64, c_int, P]),
    "snws_comm_destroy": (c_int, [P]),
    "snws_ce_blocks": (c_int, [c_int64]),
    "snws_ce_fwd_bwd": (c_int, [P, c_int, c_int64, P, c_int64, c_int, c_int, P, P, P, c_int64, c_int, P]),
    "snws_ce_fwd_bwd_w": (c_int, [P, c_int, c_int64, P, c_int64, c_int, c_int, P, P, P, P, c_int64, c_int, P]),
    "snws_argmax_stitch": (c_int, [P, c_int, c_int64, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
}

# entry points that return a status code (the others return a value)
_STATUS = {k for k, (res, _) in _SIGNATURES.items() if res is c_int} - {
    "snws_version", "snws_conv_out_size", "snws_conv_mfma_tiles", "snws_sn_stats_tiles", "snws_ce_blocks",
    "snws_upconv_stat_tiles", "snws_upconv_fwd_mode", "snws_upconv_bwd_mode", "snws_tile_minmax_blocks", "snws_conv_f32_tiles", "snws_sn_head_tiles", "snws_gemm1x1_supported", "snws_gemm1x1_stat_tiles", "snws_ocr_tiles", "snws_ocr32_runs"}

EXPORTS = tuple(_SIGNATURES)

_lib = None


class SnwsUnsupported(NotImplementedError):
    """The MFMA fast path does not take this shape; the caller may pick the direct kernels instead."""


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libsnws.so not found at %s - build it with __graft_entry__.build() "
                               "(there is no CPU fallback)" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def call(name, *args):
    """Call a status-returning entry point; translate a failure into the matching Python exception."""
    rc = getattr(lib(), name)(*args)
    if name in _STATUS and rc != 0:
        msg = lib().snws_last_error().decode("utf-8", "replace")
        if rc == ERR_INVALID:
            raise ValueError(msg)
        if rc == ERR_UNSUPPORTED:
            raise SnwsUnsupported(msg)
        raise RuntimeError(msg)
    return rc
