"""ctypes binding of the C-ABI in include/feather_hip/feather_hip.h (one declaration per exported symbol)."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STAGE_NAMES = ("wino_input", "wino_gemm", "wino_output", "igemm", "depthwise", "init", "wino_chain")


class fhip_conv_param(ctypes.Structure):
    """fhip_conv_param == booster::ConvParam field for field (reference booster.h:59-77)."""
    _fields_ = [(n, ctypes.c_int) for n in (
        "output_channels", "input_channels", "input_h", "input_w", "kernel_h", "kernel_w", "output_h", "output_w",
        "stride_h", "stride_w", "pad_left", "pad_bottom", "pad_right", "pad_top", "group", "bias_term", "activation")]


class fhip_winograd_plan(ctypes.Structure):
    _fields_ = [("tiles_x", ctypes.c_int), ("tiles_y", ctypes.c_int), ("tiles_per_image", ctypes.c_int),
                ("columns", ctypes.c_int), ("columns_padded", ctypes.c_int), ("column_block", ctypes.c_int), ("frequency_points", ctypes.c_int), ("tile_outputs", ctypes.c_int),
                ("in_channels_padded", ctypes.c_int),
                ("out_channels_padded", ctypes.c_int), ("v_offset_bytes", ctypes.c_size_t), ("v_bytes", ctypes.c_size_t),
                ("m_offset_bytes", ctypes.c_size_t), ("m_bytes", ctypes.c_size_t), ("u_bytes", ctypes.c_size_t)]


class fhip_pool_param(ctypes.Structure):
    """fhip_pool_param (feather_net.h), the fields PoolingLayer::LoadParam reads (reference pooling_layer.h:90-107)."""
    _fields_ = [(n, ctypes.c_int) for n in (
        "channels", "input_h", "input_w", "kernel_h", "kernel_w", "stride_h", "stride_w", "pad_left", "pad_right",
        "pad_top", "pad_bottom", "pooling_type", "global_pooling")]


_P = ctypes.POINTER(fhip_conv_param)
_Q = ctypes.POINTER(fhip_pool_param)
_PI = ctypes.POINTER(ctypes.c_int)
_SZ = ctypes.c_size_t
_V = ctypes.c_void_p
_I = ctypes.c_int

# symbol -> (restype, argtypes); tests/test_boundary.py checks this table against the header.
SIGNATURES = {
    "fhip_conv_assign_output_dim": (_I, [_P]),
    "fhip_conv_flops": (ctypes.c_double, [_P]),
    "fhip_conv_select_algo": (_I, [_P, ctypes.POINTER(_I)]),
    "fhip_conv_select_algo_tuned": (_I, [_P, ctypes.POINTER(_I)]),
    "fhip_conv_get_buffer_size": (_I, [_P, _I, _I, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "fhip_conv_init": (_I, [_P, _I, _V, _V, _V]),
    "fhip_conv_forward": (_I, [_P, _I, _I, _V, _V, _V, _V, _V, _V]),
    "fhip_winograd_f63_plan": (_I, [_P, _I, ctypes.POINTER(fhip_winograd_plan)]),
    "fhip_winograd_f63_transform_kernel": (_I, [_P, _V, _V, _V]),
    "fhip_winograd_f63_input_transform": (_I, [_P, _I, _V, _V, _V]),
    "fhip_winograd_f63_tile_gemm": (_I, [_P, _I, _V, _V, _V, _V]),
    "fhip_winograd_f63_output_transform": (_I, [_P, _I, _V, _V, _V, _V]),
    "fhip_stage_timing_enable": (_I, [_I]),
    "fhip_stage_timing_collect": (_I, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]),
    "fhip_last_error": (ctypes.c_char_p, []),
    "fhip_version": (ctypes.c_char_p, []),
    "fhip_device_info": (_I, [ctypes.c_char_p, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "fhip_conv_streams_1x1": (_I, [_P, _I, _I]),
    "fhip_calibrate_mfma_f32": (_I, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _V]),
    # include/feather_hip/feather_net.h -- layers between the convolutions
    "fhip_relu": (_I, [_V, _V, _SZ, _V]),
    "fhip_add": (_I, [_V, _V, _V, _SZ, _I, _V]),
    "fhip_affine": (_I, [_V, _V, _V, _V, _I, _I, _I, _I, _V]),
    "fhip_conv_can_fuse_residual": (_I, [_P, _I]),
    "fhip_conv_forward_residual": (_I, [_P, _I, _I, _V, _V, _V, _V, _V, _V, _V]),
    "fhip_conv_can_fuse_dw_pw": (_I, [_P, _P, _I]),
    "fhip_conv_forward_dw_pw": (_I, [_P, _P, _I, _V, _V, _V, _V, _V, _V, _V]),
    "fhip_conv_can_chain_winograd": (_I, [_P, _I, _P, _I, _I]),
    "fhip_conv_forward_chained": (_I, [_P, _I, _V, _V, _V, _V, _V, _V, _P, _V, _I, _V]),
    "fhip_winograd_f63_output_to_next_input": (_I, [_P, _P, _I, _V, _V, _V, _I, _V]),
    "fhip_conv_can_fuse_siblings": (_I, [_P, _I, _P, _I, _I]),
    "fhip_conv_siblings_geometry": (_I, [_P, _P, _P]),
    "fhip_conv_forward_siblings": (_I, [_P, _P, _I, _V, _V, _V, _V, _V, _V]),
    "fhip_conv_can_fuse_first_winograd": (_I, [_P, _P, _I, _I]),
    "fhip_winograd_f63_input_from_first": (_I, [_P, _P, _I, _V, _V, _V, _V, _V]),
    "fhip_conv_can_fuse_maxpool2": (_I, [_P, _I]),
    "fhip_conv_forward_maxpool2": (_I, [_P, _I, _I, _V, _V, _V, _V, _V, _V]),
    "fhip_pooling_output_dim": (_I, [_Q, _PI, _PI]),
    "fhip_pooling": (_I, [_Q, _I, _V, _V, _V]),
    "fhip_softmax": (_I, [_V, _V, _I, _I, _V]),
    # include/feather_hip/feather_net.h -- feather::Net on device blobs
    "fhip_net_create": (_I, [ctypes.POINTER(_V)]),
    "fhip_net_destroy": (_I, [_V]),
    "fhip_net_set_stream": (_I, [_V, _V]),
    "fhip_net_set_fusion": (_I, [_V, _I]),
    "fhip_net_set_graph": (_I, [_V, _I]),
    "fhip_net_set_tuned_selection": (_I, [_V, _I]),
    "fhip_net_set_concurrency": (_I, [_V, _I]),
    "fhip_net_set_sub_batches": (_I, [_V, _I]),
    "fhip_net_load_param": (_I, [_V, ctypes.c_char_p]),
    "fhip_net_load_param_mem": (_I, [_V, ctypes.c_char_p, _SZ]),
    "fhip_net_load_weights": (_I, [_V, ctypes.c_char_p]),
    "fhip_net_load_weights_mem": (_I, [_V, _V, _SZ]),
    "fhip_net_load_weights_device": (_I, [_V, _V, _SZ]),
    "fhip_net_feed_input": (_I, [_V, ctypes.c_char_p, _I, _I, _I, _I, _V, _I]),
    "fhip_net_forward": (_I, [_V]),
    "fhip_net_extract": (_I, [_V, ctypes.c_char_p, ctypes.POINTER(_V), _PI, _PI, _PI, _PI]),
    "fhip_net_extract_host": (_I, [_V, ctypes.c_char_p, _V, _SZ]),
    "fhip_net_layer_count": (_I, [_V]),
    "fhip_net_layer_info": (_I, [_V, _I, ctypes.c_char_p, ctypes.c_char_p, _I, _PI]),
    "fhip_net_layer_conv_param": (_I, [_V, _I, _P, _PI]),
    "fhip_net_layer_fused_pointwise": (_I, [_V, _I, _P, _PI]),
    "fhip_net_layer_chain": (_I, [_V, _I, _PI, _PI]),
    "fhip_net_layer_sibling": (_I, [_V, _I, _PI]),
    "fhip_net_layer_residual": (_I, [_V, _I, _PI]),
    "fhip_net_forward_timed": (_I, [_V, ctypes.POINTER(ctypes.c_float)]),
    "fhip_net_memory": (_I, [_V, ctypes.POINTER(_SZ), ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)]),
}


def lib_path() -> str:
    return os.environ.get("FEATHER_HIP_LIB", os.path.join(_HERE, "libfeather_hip.so"))


def load_library():
    """Load libfeather_hip.so.  Fails loudly: there is no fallback implementation."""
    global _LIB
    if _LIB is None:
        # PyTorch bundles its own HIP/HSA runtime; it must be the one already mapped when our library (linked against
        # libamdhip64.so.7 by SONAME) is loaded, or the process ends up with two HSA runtimes and no visible device.
        import torch  # noqa: F401  (device memory + stream provider of this host mirror)
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"feathercnn_amd: HIP library {path} is missing -- run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (or `make -C feathercnn_amd/csrc`). There is no CPU fallback.")
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB
