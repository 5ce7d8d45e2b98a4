"""ctypes front-end of oracle/extra_lib.cpp (small_mining_match, dynamic_anchor_routing) — oracle only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_extra.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "extra_lib.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_uniform.restype = ctypes.c_double
        _lib.oracle_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def small_mining_match(ov, neg_low, neg_high, pos_thres, min_match, stop_pos_thres):
    """custom_op.small_mining_match(overlaps, ...) -> (match_indices int32 [A], match_scores f32 [A])."""
    ov = np.ascontiguousarray(ov, np.float32)
    A, G = ov.shape
    idx = np.empty(A, np.int32)
    sc = np.empty(A, np.float32)
    lib().oracle_small_mining_match(_p(ov, ctypes.c_float), ctypes.c_int32(A), ctypes.c_int32(G), ctypes.c_float(neg_low),
                                    ctypes.c_float(neg_high), ctypes.c_float(pos_thres), ctypes.c_int32(min_match),
                                    ctypes.c_float(stop_pos_thres), _p(idx, ctypes.c_int32), _p(sc, ctypes.c_float))
    return idx, sc


def dynamic_anchor_routing(anchors, gt_targets, labels, mask_in, feat_h, feat_w, depth, stride, img_h, img_w,
                           training, thres, ignore_thres, u=None):
    """custom_op.dynamic_anchor_routing(...) -> (mask_out int32 [N], decode_out f32 [N,4]).
    img_h/img_w are accepted (op signature) but unused by the reference kernel.  Train mode needs `u` [N] f64."""
    anchors = np.ascontiguousarray(anchors, np.float32)
    gt_targets = np.ascontiguousarray(gt_targets, np.float32)
    labels = np.ascontiguousarray(labels, np.float32)
    mask_in = np.ascontiguousarray(mask_in, np.int32)
    N = labels.shape[0]
    mo = np.empty(N, np.int32)
    do = np.empty((N, 4), np.float32)
    if training:
        u = np.ascontiguousarray(u, np.float64)
        lib().oracle_dynamic_anchor_routing_train(_p(anchors, ctypes.c_float), _p(gt_targets, ctypes.c_float),
                                                  _p(labels, ctypes.c_float), _p(mask_in, ctypes.c_int32),
                                                  _p(u, ctypes.c_double), ctypes.c_int64(N), ctypes.c_int32(feat_h),
                                                  ctypes.c_int32(feat_w), ctypes.c_int32(depth), ctypes.c_int32(stride),
                                                  ctypes.c_float(thres), ctypes.c_float(ignore_thres),
                                                  _p(mo, ctypes.c_int32), _p(do, ctypes.c_float))
    else:
        lib().oracle_dynamic_anchor_routing_eval(_p(anchors, ctypes.c_float), _p(gt_targets, ctypes.c_float),
                                                 _p(labels, ctypes.c_float), _p(mask_in, ctypes.c_int32), ctypes.c_int64(N),
                                                 ctypes.c_int32(feat_h), ctypes.c_int32(feat_w), ctypes.c_int32(depth),
                                                 ctypes.c_int32(stride), _p(mo, ctypes.c_int32), _p(do, ctypes.c_float))
    return mo, do


def uniform_stream(seed, start, n):
    L = lib()
    return np.asarray([L.oracle_uniform(seed, start + i) for i in range(n)], np.float64)
