"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product (fealess_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("class_idx", "<i4"),
                        ("template_id", "<i4")])


class OrcBank(C.Structure):
    _fields_ = [("n_pyramids", C.c_int32), ("levels", C.c_int32), ("modalities", C.c_int32),
                ("templates", C.c_void_p), ("features", C.c_void_p)]


class OrcIcpResult(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("T", C.c_float * 3), ("dist_mean", C.c_float), ("px_ratio", C.c_float),
                ("iters", C.c_int32), ("n_corr_last", C.c_int32)]


class OrcDetectionResult(C.Structure):
    _fields_ = [("R_final", C.c_float * 9), ("T_final", C.c_float * 3), ("icp", OrcIcpResult), ("n_points", C.c_int32)]


class OrcMatch(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("similarity", C.c_float), ("class_idx", C.c_int32),
                ("template_id", C.c_int32)]


class OrcRecognitionResult(C.Structure):
    _fields_ = [("found", C.c_int32), ("best", OrcMatch), ("pose", C.c_float * 16), ("det", OrcDetectionResult),
                ("n_matches", C.c_int32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_lm_label_stride.restype = C.c_size_t
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def similarity_lut():
    out = np.zeros(256, np.uint8)
    lib().orc_similarity_lut(_p(out))
    return out


def normal_lut():
    out = np.zeros(8000, np.uint8)
    lib().orc_normal_lut(_p(out))
    return out


def spread(q, T):
    q = np.ascontiguousarray(q, np.uint8)
    out = np.zeros_like(q)
    lib().orc_spread(_p(q), q.shape[1], q.shape[0], T, _p(out))
    return out


def response_maps(s):
    s = np.ascontiguousarray(s, np.uint8)
    out = np.zeros((8,) + s.shape, np.uint8)
    lib().orc_response_maps(_p(s), s.shape[1], s.shape[0], _p(out))
    return out


def build_linear_memories(q, T):
    q = np.ascontiguousarray(q, np.uint8)
    h, w = q.shape
    stride = lib().orc_lm_label_stride(w, h, T)
    out = np.zeros(8 * stride, np.uint8)
    rc = lib().orc_build_linear_memories(_p(q), w, h, T, _p(out))
    if rc:
        raise AssertionError("reference CV_Assert")
    return out.reshape(8, stride)


def _banks(banks):
    """banks: list of fealess_amd.bank.TemplateBank (sorted by class id) -> (OrcBank array, keepalive)"""
    arr = (OrcBank * len(banks))()
    keep = []
    for i, b in enumerate(banks):
        t, f, p = b.arrays()
        keep.append((t, f, p))
        arr[i] = OrcBank(b.n_pyramids, b.levels, b.modalities, t.ctypes.data, f.ctypes.data)
    return arr, keep


def total_similarity(lms, bank, pyramid, w, h, T):
    """lms: list over modalities of (8, stride) arrays of the coarsest level."""
    arr, keep = _banks([bank])
    ptrs = (C.c_void_p * len(lms))(*[l.ctypes.data for l in lms])
    out = np.zeros((h // T, w // T), np.uint16)
    rc = lib().orc_total_similarity(ptrs, C.byref(arr[0]), pyramid, w, h, T, _p(out))
    if rc:
        raise AssertionError("reference CV_Assert")
    return out


def match_quantized(quantized, w0, h0, T_pyramid, banks, threshold, cap=1 << 20):
    qs = [np.ascontiguousarray(q, np.uint8) for q in quantized]
    levels = len(T_pyramid)
    M = len(qs) // levels
    ptrs = (C.c_void_p * len(qs))(*[q.ctypes.data for q in qs])
    T = (C.c_int * levels)(*T_pyramid)
    arr, keep = _banks(banks)
    out = np.zeros(cap, MATCH_DTYPE)
    nt = C.c_int(0)
    n = lib().orc_match_quantized(ptrs, w0, h0, levels, M, T, arr, len(banks), C.c_float(threshold), _p(out), cap,
                                  C.byref(nt))
    if n < 0:
        raise AssertionError("reference CV_Assert")
    return out[:n], nt.value


def quantized_normals(depth, distance_threshold=2000, difference_threshold=50):
    d = np.ascontiguousarray(depth, np.uint16)
    out = np.zeros(d.shape, np.uint8)
    lib().orc_quantized_normals(_p(d), d.shape[1], d.shape[0], distance_threshold, difference_threshold, _p(out))
    return out


def quantized_orientations(bgr, weak_threshold=10.0):
    b = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros(b.shape[:2], np.uint8)
    lib().orc_quantized_orientations(_p(b), b.shape[1], b.shape[0], C.c_float(weak_threshold), _p(out), None)
    return out


def pyrdown_bgr(bgr):
    b = np.ascontiguousarray(bgr, np.uint8)
    h, w = b.shape[:2]
    out = np.zeros((h // 2, w // 2, 3), np.uint8)
    lib().orc_pyrdown_bgr(_p(b), w, h, _p(out))
    return out


def resize_nn_half(q):
    q = np.ascontiguousarray(q, np.uint8)
    h, w = q.shape
    out = np.zeros((h // 2, w // 2), np.uint8)
    lib().orc_resize_nn_half(_p(q), w, h, _p(out))
    return out


def resize_linear_u8(img, dw, dh):
    a = np.ascontiguousarray(img, np.uint8)
    sh, sw = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    out = np.zeros((dh, dw) + (() if a.ndim == 2 else (cn,)), np.uint8)
    lib().orc_resize_linear_u8(_p(a), sw, sh, cn, _p(out), dw, dh)
    return out


def resize_linear_u16(img, dw, dh):
    a = np.ascontiguousarray(img, np.uint16)
    sh, sw = a.shape
    out = np.zeros((dh, dw), np.uint16)
    lib().orc_resize_linear_u16(_p(a), sw, sh, _p(out), dw, dh)
    return out


def erode_rect(mask, iterations):
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros_like(m)
    lib().orc_erode_rect(_p(m), m.shape[1], m.shape[0], iterations, _p(out))
    return out


def distance_transform_c3(img):
    m = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(m.shape, np.float32)
    lib().orc_distance_transform_c3(_p(m), m.shape[1], m.shape[0], _p(out))
    return out


FEAT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("label", "<i4")])
TEMPL_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset_x", "<i4"), ("offset_y", "<i4"),
                        ("pyramid_level", "<i4"), ("feat_begin", "<i4"), ("feat_count", "<i4")])


def extract_template_color(quantized, magnitude, mask, strong_threshold, num_features):
    q = np.ascontiguousarray(quantized, np.uint8)
    mg = np.ascontiguousarray(magnitude, np.float32)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    out = np.zeros(num_features, FEAT_DTYPE)
    ok = lib().orc_extract_template_color(_p(q), _p(mg), None if mk is None else _p(mk), q.shape[1], q.shape[0],
                                          C.c_float(strong_threshold), num_features, _p(out))
    return out if ok else None


def extract_template_depth(normal, mask, extract_threshold, num_features):
    q = np.ascontiguousarray(normal, np.uint8)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    out = np.zeros(num_features, FEAT_DTYPE)
    ok = lib().orc_extract_template_depth(_p(q), None if mk is None else _p(mk), q.shape[1], q.shape[0], extract_threshold,
                                          num_features, _p(out))
    return out if ok else None


def quantized_orientations_mag(bgr, weak_threshold=10.0):
    b = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros(b.shape[:2], np.uint8)
    mag = np.zeros(b.shape[:2], np.float32)
    lib().orc_quantized_orientations(_p(b), b.shape[1], b.shape[0], C.c_float(weak_threshold), _p(out), _p(mag))
    return out, mag


def add_template(bgr, depth, mask, levels):
    """Detector::addTemplate: returns (templates[levels*2] TEMPL_DTYPE, features list per template, bb) or None."""
    b = np.ascontiguousarray(bgr, np.uint8)
    d = np.ascontiguousarray(depth, np.uint16)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    h, w = d.shape
    t = np.zeros(levels * 2, TEMPL_DTYPE)
    f = np.zeros(levels * 2 * 63, FEAT_DTYPE)
    bb = (C.c_int * 4)()
    rc = lib().orc_add_template(_p(b), _p(d), None if mk is None else _p(mk), w, h, levels, _p(t), _p(f), bb)
    if rc:
        return None
    feats = [f[t[k]["feat_begin"]:t[k]["feat_begin"] + t[k]["feat_count"]].copy() for k in range(levels * 2)]
    return t, feats, tuple(bb)


def quantize_pyramid(bgr, depth, levels):
    """The two default modalities' quantized images per level, order [l*2 + m]."""
    out = []
    src = np.ascontiguousarray(bgr, np.uint8)
    qn = quantized_normals(depth)
    for l in range(levels):
        if l > 0:
            src = pyrdown_bgr(src)
            qn = resize_nn_half(qn)
        out.append(quantized_orientations(src))
        out.append(qn)
    return out


def match_images(bgr, depth, T_pyramid, banks, threshold, cap=1 << 20, masks=None):
    b = np.ascontiguousarray(bgr, np.uint8)
    d = np.ascontiguousarray(depth, np.uint16)
    h, w = d.shape
    levels = len(T_pyramid)
    T = (C.c_int * levels)(*T_pyramid)
    arr, keep = _banks(banks)
    out = np.zeros(cap, MATCH_DTYPE)
    nt = C.c_int(0)
    mks = [None, None]
    if masks is not None:
        mks = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in masks]
    n = lib().orc_match_images_masked(_p(b), _p(d), w, h, levels, T, arr, len(banks), C.c_float(threshold),
                                      None if mks[0] is None else _p(mks[0]), None if mks[1] is None else _p(mks[1]),
                                      _p(out), cap, C.byref(nt), None)
    if n < 0:
        raise AssertionError("reference CV_Assert")
    return out[:n], nt.value


def depth_to_3d(depth, fx, fy, cx, cy):
    d = np.ascontiguousarray(depth, np.uint16)
    out = np.zeros(d.shape + (3,), np.float32)
    lib().orc_depth_to_3d(_p(d), d.shape[1], d.shape[0], C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy),
                          _p(out))
    return out


def svd3(A):
    A = np.ascontiguousarray(A, np.float32)
    w = np.zeros(3, np.float32)
    u = np.zeros((3, 3), np.float32)
    vt = np.zeros((3, 3), np.float32)
    lib().orc_svd3(_p(A), _p(w), _p(u), _p(vt))
    return w, u, vt


def _icp_dict(r):
    return dict(R=np.array(r.R, np.float32).reshape(3, 3), T=np.array(r.T, np.float32), dist_mean=np.float32(r.dist_mean),
                px_ratio=np.float32(r.px_ratio), iters=int(r.iters), n_corr_last=int(r.n_corr_last))


def icp(ref, model, icp_it_thr=4, dist_mean_thr=0.0, dist_diff_thr=0.0, accum64=False, use_kdtree=True, trace=False):
    ref = np.ascontiguousarray(ref, np.float32).reshape(-1, 3)
    model = np.ascontiguousarray(model, np.float32).reshape(-1, 3)
    res = OrcIcpResult()
    tr = np.full((max(icp_it_thr, 1), 23), np.nan, np.float32)
    rc = lib().orc_icp(_p(ref), len(ref), _p(model), len(model), icp_it_thr, C.c_float(dist_mean_thr),
                       C.c_float(dist_diff_thr), int(accum64), int(use_kdtree), C.byref(res),
                       _p(tr) if trace else None, icp_it_thr if trace else 0)
    d = _icp_dict(res)
    d["rc"] = rc
    if trace:
        d["trace"] = tr
    return d


def detection(model_depth_mm, scene_depth_mm, K, rect_model, rect_ref, icp_it_thr, dist_mean_thr, dist_diff_thr,
              r_match, t_match, accum64=False, use_kdtree=True):
    md = np.ascontiguousarray(model_depth_mm, np.uint16)
    sd = np.ascontiguousarray(scene_depth_mm, np.uint16)
    h, w = sd.shape
    rm = (C.c_int * 4)(*[int(v) for v in rect_model])
    rr = (C.c_int * 4)(*[int(v) for v in rect_ref])
    rmat = (C.c_float * 9)(*np.asarray(r_match, np.float32).ravel())
    tvec = (C.c_float * 3)(*np.asarray(t_match, np.float32).ravel())
    res = OrcDetectionResult()
    rc = lib().orc_detection(_p(md), _p(sd), w, h, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
                             rm, rr, icp_it_thr, C.c_float(dist_mean_thr), C.c_float(dist_diff_thr), rmat, tvec,
                             int(accum64), int(use_kdtree), C.byref(res))
    return dict(rc=rc, R_final=np.array(res.R_final, np.float32).reshape(3, 3), T_final=np.array(res.T_final, np.float32),
                icp=_icp_dict(res.icp), n_points=int(res.n_points))


def recognition(bgr, depth, K, T_pyramid, bank, threshold=75.0, icp_it_thr=10, dist_mean_thr=0.5, dist_diff_thr=0.01,
                accum64=False, use_kdtree=True):
    b = np.ascontiguousarray(bgr, np.uint8)
    d = np.ascontiguousarray(depth, np.uint16)
    h, w = d.shape
    levels = len(T_pyramid)
    T = (C.c_int * levels)(*T_pyramid)
    arr, keep = _banks([bank])
    t, f, p = keep[0]
    zero = np.zeros((h, w), np.uint16)    # pyramids without a depth render (bank.py): an empty render, as the product uploads
    mds = [zero if m is None else np.ascontiguousarray(m, np.uint16) for m in bank.model_depths]
    mds = mds + [zero] * (bank.n_pyramids - len(mds))
    mptr = (C.c_void_p * len(mds))(*[m.ctypes.data for m in mds])
    res = OrcRecognitionResult()
    rc = lib().orc_recognition(_p(b), _p(d), w, h, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
                               levels, T, C.byref(arr[0]), _p(p), mptr, C.c_float(threshold), icp_it_thr,
                               C.c_float(dist_mean_thr), C.c_float(dist_diff_thr), int(accum64), int(use_kdtree),
                               C.byref(res))
    return dict(rc=rc, found=int(res.found), n_matches=int(res.n_matches),
                best=dict(x=res.best.x, y=res.best.y, similarity=np.float32(res.best.similarity),
                          class_idx=res.best.class_idx, template_id=res.best.template_id),
                pose=np.array(res.pose, np.float32).reshape(4, 4),
                det=dict(R_final=np.array(res.det.R_final, np.float32).reshape(3, 3),
                         T_final=np.array(res.det.T_final, np.float32), icp=_icp_dict(res.det.icp),
                         n_points=int(res.det.n_points)))


def last_stage_ms():
    """(linemod_ms, icp_ms) of the calling thread's last recognition(): the reference's own timer points
    ("Time of linemod" / "Time of ICP", CadReco/obj_reco_lmicp.cpp:125,202)."""
    out = (C.c_double * 2)()
    lib().orc_last_stage_ms(out)
    return float(out[0]), float(out[1])


def _reco_dict(res, rc=0):
    return dict(rc=rc, found=int(res.found), n_matches=int(res.n_matches),
                best=dict(x=res.best.x, y=res.best.y, similarity=np.float32(res.best.similarity),
                          class_idx=res.best.class_idx, template_id=res.best.template_id),
                pose=np.array(res.pose, np.float32).reshape(4, 4),
                det=dict(R_final=np.array(res.det.R_final, np.float32).reshape(3, 3),
                         T_final=np.array(res.det.T_final, np.float32), icp=_icp_dict(res.det.icp),
                         n_points=int(res.det.n_points)))


def recognition_topk(bgr, depth, K, T_pyramid, bank, k, threshold=75.0, icp_it_thr=10, dist_mean_thr=0.5, dist_diff_thr=0.01,
                     nms_dist=None):
    """First k matches refined like Recognition() does for matches[0]; with nms_dist also the NMS winners."""
    b = np.ascontiguousarray(bgr, np.uint8)
    d = np.ascontiguousarray(depth, np.uint16)
    h, w = d.shape
    levels = len(T_pyramid)
    T = (C.c_int * levels)(*T_pyramid)
    arr, keep = _banks([bank])
    t, f, p = keep[0]
    zero = np.zeros((h, w), np.uint16)    # pyramids without a depth render (bank.py): an empty render, as the product uploads
    mds = [zero if m is None else np.ascontiguousarray(m, np.uint16) for m in bank.model_depths]
    mds = mds + [zero] * (bank.n_pyramids - len(mds))
    mptr = (C.c_void_p * len(mds))(*[m.ctypes.data for m in mds])
    res = (OrcRecognitionResult * k)()
    n = lib().orc_recognition_topk(_p(b), _p(d), w, h, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
                                   levels, T, C.byref(arr[0]), _p(p), mptr, C.c_float(threshold), icp_it_thr,
                                   C.c_float(dist_mean_thr), C.c_float(dist_diff_thr), 0, 1, k, res)
    if n < 0:
        raise AssertionError("reference CV_Assert")
    out = [_reco_dict(res[i]) for i in range(n)]
    if nms_dist is None:
        return out
    win = (C.c_int * max(1, n))()
    nw = lib().orc_nms(res, n, C.c_float(nms_dist), win)
    return out, [int(win[i]) for i in range(nw)]


def gaussian7_bgr(bgr):
    b = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros_like(b)
    lib().orc_gaussian7_bgr(_p(b), b.shape[1], b.shape[0], _p(out))
    return out


def median5(img):
    m = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(m)
    lib().orc_median5(_p(m), m.shape[1], m.shape[0], _p(out))
    return out
