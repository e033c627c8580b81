"""ctypes binding of libfealess_hip.so (the C ABI declared in include/fealess_hip.h).

The library is the product: hand-written HIP kernels for gfx950.  There is no CPU fallback --
if the shared object is missing this module raises, and without a HIP device
``fl_context_create`` returns FL_ERR_NO_DEVICE.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FEALESS_HIP_LIB: development knob -- another build of the same library (A/B timing of two builds on one GPU box)
LIB_PATH = os.environ.get("FEALESS_HIP_LIB") or os.path.join(_HERE, "csrc", "libfealess_hip.so")

FL_OK = 0
FL_ERR_INVALID, FL_ERR_HIP, FL_ERR_ASSERT, FL_ERR_OVERFLOW, FL_ERR_NO_DEVICE, FL_ERR_STATE, FL_ERR_NO_TEMPLATE = -1, -2, -3, -4, -5, -6, -7
FL_MEM_HOST, FL_MEM_DEVICE = 0, 1
FL_ICP_PARITY, FL_ICP_FAST, FL_ICP_POINT_TO_PLANE = 0, 1, 2


class Feature(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("label", C.c_int32)]


class Template(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("offset_x", C.c_int32), ("offset_y", C.c_int32),
                ("pyramid_level", C.c_int32), ("feat_begin", C.c_int32), ("feat_count", C.c_int32)]


class Match(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("similarity", C.c_float), ("class_idx", C.c_int32),
                ("template_id", C.c_int32)]


class Intrinsics(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("T", C.c_float * 3), ("dist_mean", C.c_float), ("px_ratio", C.c_float),
                ("iters", C.c_int32), ("n_corr_last", C.c_int32)]


class DetectionResult(C.Structure):
    _fields_ = [("R_final", C.c_float * 9), ("T_final", C.c_float * 3), ("icp", IcpResult), ("n_points", C.c_int32),
                ("status", C.c_int32)]


class RecognitionResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("found", C.c_int32), ("n_matches", C.c_int32), ("best", Match),
                ("pose", C.c_float * 16), ("det", DetectionResult)]


class RecognitionParams(C.Structure):
    _fields_ = [("matching_threshold", C.c_float), ("icp_it_thr", C.c_int32), ("dist_mean_thr", C.c_float),
                ("dist_diff_thr", C.c_float), ("icp_mode", C.c_int32)]


class StageTimes(C.Structure):
    _fields_ = [("frontend_ms", C.c_float), ("linmem_ms", C.c_float), ("scan_ms", C.c_float), ("refine_ms", C.c_float),
                ("sort_ms", C.c_float), ("backproject_ms", C.c_float), ("icp_ms", C.c_float), ("total_ms", C.c_float),
                ("icp_iters_total", C.c_int32), ("icp_launches", C.c_int32), ("scan_algorithmic_bytes", C.c_double),
                ("lazy_frontend_ms", C.c_float), ("reserved0", C.c_float)]


FL_TOPK_OVERFLOW = -2      # fl_export_topk_batch: template id of record 0 of a frame whose candidate buffers overflowed

_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_D = C.c_double

# name -> (restype, argtypes); every symbol include/fealess_hip.h declares
SIGNATURES = {
    "fl_abi_version": (_I, []),
    "fl_context_create": (_I, [_I, C.POINTER(_P)]),
    "fl_context_destroy": (None, [_P]),
    "fl_last_error": (C.c_char_p, [_P]),
    "fl_context_set_stream": (_I, [_P, _P]),
    "fl_context_synchronize": (_I, [_P]),
    "fl_context_get_stream": (_P, [_P]),
    "fl_context_get_device": (_I, [_P]),
    "fl_detector_get_context": (_P, [_P]),
    "fl_context_set_option": (_I, [_P, C.c_char_p, C.c_long]),
    "fl_context_get_option": (_I, [_P, C.c_char_p, C.POINTER(C.c_long)]),
    "fl_detector_create": (_I, [_P, _I, _I, C.POINTER(_I), C.POINTER(_P)]),
    "fl_detector_destroy": (None, [_P]),
    "fl_detector_add_class": (_I, [_P, C.c_char_p, _I, _P, _P, _I, _P]),
    "fl_detector_set_model_depths": (_I, [_P, _I, _I, _I, _P, _I, _I, _I]),
    "fl_detector_finalize": (_I, [_P, _I, _I, _I, _I]),
    "fl_detector_grow_candidates": (_I, [_P, _I, C.POINTER(_I)]),
    "fl_detector_set_class_filter": (_I, [_P, C.POINTER(C.c_char_p), _I]),
    "fl_detector_num_templates": (_I, [_P]),
    "fl_detector_num_classes": (_I, [_P]),
    "fl_quantized_orientations": (_I, [_P, _P, _I, _I, _F, _P, _I]),
    "fl_quantized_normals": (_I, [_P, _P, _I, _I, _I, _I, _P, _I]),
    "fl_pyrdown_bgr": (_I, [_P, _P, _I, _I, _P, _I]),
    "fl_extract_template_pyramid": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, C.POINTER(_I)]),
    "fl_resize_linear_bgr8": (_I, [_P, _P, _I, _I, _P, _I, _I, _I]),
    "fl_resize_linear_u16": (_I, [_P, _P, _I, _I, _P, _I, _I, _I]),
    "fl_lm_label_stride": (C.c_size_t, [_I, _I, _I]),
    "fl_build_linear_memories": (_I, [_P, _P, _I, _I, _I, _P, _I]),
    "fl_depth_to_3d": (_I, [_P, _P, _I, _I, _D, _D, _D, _D, _P, _I]),
    "fl_icp": (_I, [_P, _P, _I, _P, _I, _I, _F, _F, _I, _I, C.POINTER(IcpResult)]),
    "fl_icp_point_to_plane": (_I, [_P, _P, _P, _I, _P, _I, _I, _F, _F, _I, C.POINTER(IcpResult)]),
    "fl_detection": (_I, [_P, _P, _P, _I, _I, C.POINTER(Intrinsics), C.POINTER(_I), C.POINTER(_I), _I, _F, _F,
                          C.POINTER(_F), C.POINTER(_F), _I, _I, C.POINTER(DetectionResult)]),
    "fl_match_quantized": (_I, [_P, C.POINTER(_P), _I, _F, _P, _I, C.POINTER(_I)]),
    "fl_match_frame": (_I, [_P, _P, _P, _I, _F, _P, _I, C.POINTER(_I)]),
    "fl_match_frame_masked": (_I, [_P, _P, _P, C.POINTER(_P), _I, _F, _P, _I, C.POINTER(_I)]),
    "fl_match_batch_submit": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P), _I, _F]),
    "fl_match_batch_collect": (_I, [_P, _I, _P, _I, C.POINTER(_I)]),
    "fl_similarity_maps": (_I, [_P, _I, _I, _P]),
    "fl_last_quantized": (_I, [_P, _P]),
    "fl_recognize_batch": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P), _I, C.POINTER(Intrinsics),
                                C.POINTER(RecognitionParams), _P]),
    "fl_recognize_batch_zoom": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P), _I, _I, _I, C.POINTER(Intrinsics),
                                     C.POINTER(RecognitionParams), _P]),
    "fl_recognize_submit": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P), _I, C.POINTER(Intrinsics),
                                 C.POINTER(RecognitionParams)]),
    "fl_recognize_collect": (_I, [_P, _I, _P]),
    "fl_recognize_topk": (_I, [_P, _P, _P, _I, C.POINTER(Intrinsics), C.POINTER(RecognitionParams), _I, _P, C.POINTER(_I)]),
    "fl_recognize_batch_topk": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P), _I, C.POINTER(Intrinsics), C.POINTER(RecognitionParams), _I, _P,
                                     C.POINTER(_I)]),
    "fl_nms": (_I, [_P, _I, _F, C.POINTER(_I), C.POINTER(_I)]),
    "fl_export_topk": (_I, [_P, _I, _I, _I, _P]),
    "fl_merge_topk": (_I, [_P, _I, _P, _I]),
    "fl_export_topk_batch": (_I, [_P, _I, _I, _I, _P]),
    "fl_merge_topk_batch": (_I, [_P, _I, _I, _I, _P, _I, C.POINTER(_I)]),
    "fl_refine_matches": (_I, [_P, _I, C.POINTER(C.c_int32), _P, C.POINTER(Intrinsics), C.POINTER(RecognitionParams), _P]),
    "fl_select_best_batch": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fl_refine_selected": (_I, [_P, _I, C.POINTER(Intrinsics), C.POINTER(RecognitionParams), _P, C.c_size_t, _P]),
    "fl_last_stage_times": (_I, [_P, C.POINTER(StageTimes)]),
    "fl_frame_counters": (_I, [_P, _I, C.POINTER(C.c_int32)]),
}

class MgResult(C.Structure):         # include/fealess_mg.h fl_mg_result
    _fields_ = [("status", C.c_int32), ("found", C.c_int32), ("best", Match), ("pose", C.c_float * 16)]


FL_MG_ID_BYTES = 128
MG_LIB_PATH = os.path.join(_HERE, "mg", "libfealess_mg.so")
# every symbol include/fealess_mg.h declares (libfealess_mg.so: the C++ multi-GPU host on RCCL)
MG_SIGNATURES = {
    "fl_mg_unique_id": (_I, [_P, C.c_size_t]),
    "fl_mg_create": (_I, [_P, _P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "fl_mg_destroy": (None, [_P]),
    "fl_mg_last_error": (C.c_char_p, [_P]),
    "fl_mg_recognize_batch": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P), _I, C.POINTER(Intrinsics), C.POINTER(RecognitionParams), _P]),
    "fl_mg_last_stats": (_I, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
}

_lib = None
_mg_lib = None


def load_mg():
    """Load libfealess_mg.so (links librccl and libfealess_hip.so); raises if it has not been built."""
    global _mg_lib
    if _mg_lib is None:
        load()
        if not os.path.exists(MG_LIB_PATH):
            raise RuntimeError(f"{MG_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(MG_LIB_PATH)
        for name, (res, args) in MG_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _mg_lib = lib
    return _mg_lib



def load():
    """Load libfealess_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). fealess_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
