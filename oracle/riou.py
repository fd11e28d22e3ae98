"""ctypes front-end for oracle/riou_oracle.c (this repo's restatement) and, when present,
oracle/_ref/libref_riou.so (the reference's own IoU arithmetic compiled from /root/reference).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)


def _build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _load(path, build=True):
    if not os.path.exists(path) and build:
        _build()
    return C.CDLL(path)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load(os.path.join(_HERE, "liboracle_riou.so"))
        _lib.oracle_rotate_iou.restype = C.c_float
        _lib.oracle_rotate_iou.argtypes = [_f32p, _f32p]
        _lib.oracle_rnms.restype = C.c_int
        _lib.oracle_rnms.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, _i64p, C.c_int, _i64p]
        _lib.oracle_rnms_bitmatrix.restype = C.c_int
        _lib.oracle_rnms_bitmatrix.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, _i64p, C.c_void_p]
        _lib.oracle_riou_matrix.restype = None
        _lib.oracle_riou_matrix.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p]
        _lib.oracle_riou_pairs.restype = None
        _lib.oracle_riou_pairs.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_int, _f32p]
        _lib.oracle_convert_region.restype = None
        _lib.oracle_convert_region.argtypes = [_f32p, _f32p]
        _lib.oracle_sincosf_cr.restype = None
        _lib.oracle_sincosf_cr.argtypes = [C.c_float, _f32p, _f32p]
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_riou.so"))


def ref():
    """The reference-built checker (oracle/_ref).  Raises if it was never built."""
    global _ref
    if _ref is None:
        _ref = _load(os.path.join(_HERE, "_ref", "libref_riou.so"), build=False)
        _ref.ref_rotate_iou.restype = C.c_float
        _ref.ref_rotate_iou.argtypes = [_f32p, _f32p]
        _ref.ref_rnms.restype = C.c_int
        _ref.ref_rnms.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, _i64p]
        _ref.ref_riou_matrix.restype = None
        _ref.ref_riou_matrix.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p]
        _ref.ref_convert_region.restype = None
        _ref.ref_convert_region.argtypes = [_f32p, _f32p]
    return _ref


def _rows(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 5:
        raise ValueError("boxes must be [n, >=5]")
    return a, a.ctypes.data_as(_f32p), a.shape[0], a.shape[1]


def num_threads():
    return int(lib().oracle_num_threads())


def rnms(dets, thr, nthreads=1, use_ref=False, return_pairs=False):
    """Greedy rotated NMS on the CPU.  dets [n,>=6] float32 -> kept original indices (ascending int64)."""
    a, p, n, s = _rows(dets)
    if s < 6:
        raise ValueError("dets must be [n, >=6]")
    keep = np.empty(max(n, 1), dtype=np.int64)
    kp = keep.ctypes.data_as(_i64p)
    if use_ref:
        k = ref().ref_rnms(p, n, s, float(thr), kp)
        return keep[:k].copy()
    pairs = C.c_int64(0)
    k = lib().oracle_rnms(p, n, s, float(thr), kp, int(nthreads), C.byref(pairs))
    if return_pairs:
        return keep[:k].copy(), int(pairs.value)
    return keep[:k].copy()


def rnms_bitmatrix(dets, thr, return_mask=False):
    a, p, n, s = _rows(dets)
    keep = np.empty(max(n, 1), dtype=np.int64)
    cb = (n + 63) // 64
    mask = np.zeros((max(n, 1), max(cb, 1)), dtype=np.uint64)
    k = lib().oracle_rnms_bitmatrix(p, n, s, float(thr), keep.ctypes.data_as(_i64p),
                                    mask.ctypes.data_as(C.c_void_p))
    return (keep[:k].copy(), mask) if return_mask else keep[:k].copy()


def riou_matrix(b1, b2, use_ref=False):
    a1, p1, n1, s1 = _rows(b1)
    a2, p2, n2, s2 = _rows(b2)
    out = np.empty((n1, n2), dtype=np.float32)
    fn = ref().ref_riou_matrix if use_ref else lib().oracle_riou_matrix
    fn(p1, n1, s1, p2, n2, s2, out.ctypes.data_as(_f32p))
    return out


def riou_pairs(b1, b2):
    a1, p1, n1, s1 = _rows(b1)
    a2, p2, n2, s2 = _rows(b2)
    assert n1 == n2
    out = np.empty(n1, dtype=np.float32)
    lib().oracle_riou_pairs(p1, s1, p2, s2, n1, out.ctypes.data_as(_f32p))
    return out


def corners(boxes, use_ref=False):
    a, p, n, s = _rows(boxes)
    out = np.empty((n, 8), dtype=np.float32)
    fn = ref().ref_convert_region if use_ref else lib().oracle_convert_region
    for i in range(n):
        fn(out[i].ctypes.data_as(_f32p), a[i].ctypes.data_as(_f32p))
    return out


def sincos(a):
    s = C.c_float()
    c = C.c_float()
    lib().oracle_sincosf_cr(float(np.float32(a)), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def random_boxes(n, seed=0, extent=608.0):
    """SURVEY.md section 8(d) config-3 distribution: cx,cy ~ U(0,extent); w,h = 8*16^U(0,1);
    angle ~ U(-pi/2, pi/2); scores = a random permutation of (i+0.5)/n (all distinct)."""
    rng = np.random.default_rng(seed)
    d = np.empty((n, 6), dtype=np.float32)
    d[:, 0] = rng.uniform(0, extent, n)
    d[:, 1] = rng.uniform(0, extent, n)
    d[:, 2] = 8.0 * 16.0 ** rng.uniform(0, 1, n)
    d[:, 3] = 8.0 * 16.0 ** rng.uniform(0, 1, n)
    d[:, 4] = rng.uniform(-np.pi / 2, np.pi / 2, n)
    d[:, 5] = (rng.permutation(n) + 0.5) / n
    return d
