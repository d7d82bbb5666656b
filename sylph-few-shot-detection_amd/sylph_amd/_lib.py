"""ctypes binding of libsylph_hip.so (C ABI in include/sylph_hip.h).

The library is the ONLY compute path of this package: there is no eager/CPU fallback.  If the
shared object is missing or does not load, importing this module's ``lib()`` raises.
torch must be imported first so that the HIP runtime (libamdhip64.so.7) already loaded by torch is
the one the library binds to: device pointers are then shared between torch tensors and our kernels.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_void_p

import torch  # noqa: F401  (loads the HIP runtime first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SYLPH_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libsylph_hip.so")  # override: A/B kernel builds

SYLPH_F32 = 0
SYLPH_BF16 = 1
SYLPH_F32S = 2


class SylphConfig(Structure):
    _fields_ = [
        ("resnet_depth", c_int), ("stride_in_1x1", c_int), ("num_cls_convs", c_int), ("num_box_convs", c_int),
        ("nlevels", c_int), ("strides", c_int * 8), ("pixel_mean", c_float * 3), ("pixel_std", c_float * 3),
        ("size_divisibility", c_int), ("use_scale", c_int), ("cond_use_bias", c_int),
        ("pre_nms_thresh", c_float), ("pre_nms_topk", c_int), ("nms_thresh", c_float), ("post_nms_topk", c_int),
        ("thresh_with_ctr", c_int), ("quality_mode", c_int),
        ("cg_tower_layers", c_int), ("cg_has_bias", c_int), ("cg_bias_l2_norm", c_int), ("cg_post_norm", c_int),
        ("cg_conv_l2_norm", c_int), ("cg_use_weight_scale", c_int), ("prior_prob", c_float), ("cand_cap", c_int),
        ("cg_type", c_int), ("tok_num_conv", c_int), ("tok_num_fc", c_int), ("enc_layers", c_int),
        ("head_num_fc", c_int), ("head_fc_dim", c_int), ("cg_meta_bias", c_int), ("cg_has_weight", c_int), ("cg_has_scale", c_int),
        ("num_share_convs", c_int), ("tower_norm", c_int), ("cg_tower_gn_mask", c_int), ("cg_tower_relu_mask", c_int),
    ]


# name -> (restype, argtypes); every symbol declared in include/sylph_hip.h
PROTOTYPES = {
    "sylph_config_default": (None, [POINTER(SylphConfig)]),
    "sylph_ctx_create": (c_int, [c_int, c_int, POINTER(c_void_p)]),
    "sylph_ctx_destroy": (None, [c_void_p]),
    "sylph_last_error": (c_char_p, []),
    "sylph_set_stream": (c_int, [c_void_p, c_void_p]),
    "sylph_set_config": (c_int, [c_void_p, POINTER(SylphConfig)]),
    "sylph_load_weight": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "sylph_finalize_weights": (c_int, [c_void_p]),
    "sylph_preprocess": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_int),
                                 POINTER(c_int), POINTER(c_int)]),
    "sylph_preprocess_u8": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                    POINTER(c_int), c_int, POINTER(c_int), POINTER(c_int)]),
    "sylph_export_input": (c_int, [c_void_p, c_void_p]),
    "sylph_backbone_fpn": (c_int, [c_void_p]),
    "sylph_import_pyramid": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int),
                                     POINTER(c_void_p)]),
    "sylph_export_pyramid": (c_int, [c_void_p, c_int, c_void_p]),
    "sylph_fcos_head": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "sylph_fcos_head_pretrained": (c_int, [c_void_p, POINTER(c_int)]),
    "sylph_export_head": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sylph_import_head": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sylph_roi_align": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sylph_decode_nms": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sylph_codegen": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sylph_codegen_classes": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "sylph_codegen_weight_norm": (c_int, [c_void_p, c_void_p]),
    "sylph_normalize_codes": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "sylph_reduce_codes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int]),
    "sylph_conv2d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                             c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "sylph_group_norm": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "sylph_stem_maxpool": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sylph_set_debug_taps": (c_int, [c_void_p, c_int]),
    "sylph_export_stage": (c_int, [c_void_p, c_int, c_void_p]),
    "sylph_export_tower": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sylph_bottleneck": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_void_p),
                                 POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "sylph_allgather_codes": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "sylph_comm_unique_id": (c_int, [ctypes.c_char_p]),
    "sylph_comm_init_rank": (c_int, [c_void_p, ctypes.c_char_p, c_int, c_int, POINTER(c_void_p)]),
    "sylph_comm_destroy": (c_int, [c_void_p]),
    "sylph_fpn_lateral": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sylph_device_bytes": (c_int64, [c_void_p]),
    "sylph_profile_enable": (c_int, [c_void_p, c_int]),
    "sylph_bench_conv": (c_int, [c_void_p] + [c_int] * 12 + [POINTER(c_float), POINTER(ctypes.c_double)]),
    "sylph_profile_read": (c_int, [c_void_p, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64)]),
    "sylph_profile_read_kernels": (c_int, [c_void_p, c_int, ctypes.c_char_p, POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                           POINTER(c_int64), POINTER(c_int)]),
}

_LIB = None


def lib():
    """Load libsylph_hip.so (once).  Raises if the HIP extension has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C sylph-few-shot-detection_amd/csrc).  There is no CPU fallback.")
        _LIB = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(_LIB, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
    return _LIB


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().sylph_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libsylph_hip {what}: {msg}")
