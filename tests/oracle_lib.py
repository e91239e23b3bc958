"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")
NO_SECOND = 2**31 - 1
METRIC_DOT, METRIC_L2 = 0, 1


class Params(C.Structure):
    _fields_ = [
        ("ratio_num", C.c_int32), ("ratio_den", C.c_int32), ("mutual", C.c_int32), ("reserved", C.c_int32),
        ("max_dist_sq", C.c_int64), ("min_score", C.c_float), ("max_dist_sq_f", C.c_float),
    ]


def params(ratio=(4, 5), mutual=True, max_dist_sq=-1, min_score=float("-inf"), max_dist_sq_f=-1.0):
    num, den = ratio if ratio else (0, 1)
    return Params(num, den, 1 if mutual else 0, 0, max_dist_sq, min_score, max_dist_sq_f)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("oracle not built: make -C oracle")
        _lib = C.CDLL(_PATH)
        _lib.orc_dot_f32.restype = C.c_float
        _lib.orc_sift_write.restype = C.c_long
        _lib.orc_matchtxt_record.restype = C.c_long
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


THREADS = min(8, os.cpu_count() or 1)


def knn2_u8(A, B, threads=THREADS):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    n, m = len(A), len(B)
    idx = np.empty(max(n, 1), np.int32)
    d1 = np.empty(max(n, 1), np.int32)
    d2 = np.empty(max(n, 1), np.int32)
    lib().orc_knn2_u8(_p(A), n, _p(B), m, threads, _p(idx), _p(d1), _p(d2))
    return idx[:n], d1[:n], d2[:n]


def colbest_u8(A, B, threads=THREADS):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    n, m = len(A), len(B)
    ci = np.empty(max(m, 1), np.int32)
    cd = np.empty(max(m, 1), np.int32)
    lib().orc_colbest_u8(_p(A), n, _p(B), m, threads, _p(ci), _p(cd))
    return ci[:m], cd[:m]


def match_u8(A, B, p, threads=THREADS):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    n, m = len(A), len(B)
    cap = max(n, 1)
    i0 = np.empty(cap, np.int32)
    i1 = np.empty(cap, np.int32)
    d = np.empty(cap, np.int32)
    nm = lib().orc_match_u8(_p(A), n, _p(B), m, C.byref(p), threads, _p(i0), _p(i1), _p(d), cap)
    return i0[:nm], i1[:nm], d[:nm]


def match_dense_u8(A, B, p, threads=THREADS):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    n, m = len(A), len(B)
    m0 = np.empty(max(n, 1), np.int32)
    m1 = np.empty(max(m, 1), np.int32)
    lib().orc_match_dense_u8(_p(A), n, _p(B), m, C.byref(p), threads, _p(m0), _p(m1))
    return m0[:n], m1[:m]


def knn2_f32(A, B, metric, threads=THREADS):
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    n, m, dim = len(A), len(B), A.shape[1]
    idx = np.empty(max(n, 1), np.int32)
    s1 = np.empty(max(n, 1), np.float32)
    s2 = np.empty(max(n, 1), np.float32)
    lib().orc_knn2_f32(_p(A), n, _p(B), m, dim, metric, threads, _p(idx), _p(s1), _p(s2))
    return idx[:n], s1[:n], s2[:n]


def colbest_f32(A, B, metric, threads=THREADS):
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    n, m, dim = len(A), len(B), A.shape[1]
    ci = np.empty(max(m, 1), np.int32)
    cs = np.empty(max(m, 1), np.float32)
    lib().orc_colbest_f32(_p(A), n, _p(B), m, dim, metric, threads, _p(ci), _p(cs))
    return ci[:m], cs[:m]


def match_dense_f32(A, B, metric, p, threads=THREADS):
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    n, m, dim = len(A), len(B), A.shape[1]
    m0 = np.empty(max(n, 1), np.int32)
    m1 = np.empty(max(m, 1), np.int32)
    b0 = np.empty(max(n, 1), np.float32)
    b1 = np.empty(max(m, 1), np.float32)
    lib().orc_match_dense_f32(_p(A), n, _p(B), m, dim, metric, C.byref(p), threads, _p(m0), _p(m1), _p(b0), _p(b1))
    return m0[:n], m1[:m], b0[:n], b1[:m]


def nearest_embed_f32(x, emb, threads=THREADS):
    x = np.ascontiguousarray(x, np.float32)
    emb = np.ascontiguousarray(emb, np.float32)
    b, d = x.shape[0], x.shape[1]
    npts = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    am = np.empty((b, npts), np.int64)
    res = np.empty((b, d, npts), np.float32)
    lib().orc_nearest_embed_f32(_p(x), b, d, npts, _p(emb), emb.shape[1], threads, _p(am), _p(res))
    return res.reshape(x.shape), am.reshape((b,) + tuple(x.shape[2:]))


def merge_top2_f32(k1, k2, i1, i2, metric):
    k1 = np.ascontiguousarray(k1, np.float32)
    k2 = np.ascontiguousarray(k2, np.float32)
    i1 = np.ascontiguousarray(i1, np.int32)
    i2 = np.ascontiguousarray(i2, np.int32)
    ns, nq = k1.shape
    ok1 = np.empty(nq, np.float32)
    ok2 = np.empty(nq, np.float32)
    oi1 = np.empty(nq, np.int32)
    oi2 = np.empty(nq, np.int32)
    lib().orc_merge_top2_f32(_p(k1), _p(k2), _p(i1), _p(i2), ns, nq, metric, _p(ok1), _p(ok2), _p(oi1), _p(oi2))
    return ok1, ok2, oi1, oi2


def sift_write(path, loc, des):
    loc = np.ascontiguousarray(loc, np.float32)
    des = np.ascontiguousarray(des, np.uint8)
    return lib().orc_sift_write(os.fsencode(path), _p(loc), loc.shape[1], _p(des), len(des))


def sift_read(path):
    nloc = C.c_int(0)
    n = lib().orc_sift_read(os.fsencode(path), None, None, 0, C.byref(nloc))
    if n <= 0:
        return None
    loc = np.empty((n, nloc.value), np.float32)
    des = np.empty((n, 128), np.uint8)
    lib().orc_sift_read(os.fsencode(path), _p(loc), _p(des), n, C.byref(nloc))
    return loc, des


def sift_read_ascii(path, nmax=1 << 20):
    loc = np.zeros((nmax, 5), np.float32)
    des = np.zeros((nmax, 128), np.uint8)
    fn = lib().orc_sift_read_ascii
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
    n = fn(os.fsencode(path), _p(loc), _p(des), nmax)
    k = min(n, nmax)
    return n, loc[:k].copy(), des[:k].copy()


def sift_write_ascii(path, loc, des):
    loc = np.ascontiguousarray(loc, np.float32)
    des = np.ascontiguousarray(des, np.uint8)
    fn = lib().orc_sift_write_ascii
    fn.restype = C.c_long
    fn.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
    return fn(os.fsencode(path), _p(loc), _p(des), len(des))


def sift_sort(loc, des):
    """SortSIFT on copies: (loc, des, order)."""
    loc = np.array(loc, np.float32, order="C")
    des = np.array(des, np.uint8, order="C")
    order = np.zeros(max(len(des), 1), np.int32)
    fn = lib().orc_sift_sort
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    fn(_p(loc), _p(des), len(des), _p(order))
    return loc, des, order[:len(des)]


def sift_convert_a2b(loc, des):
    """The in-memory part of ConvertA2B on copies: (converted?, loc, des)."""
    loc = np.array(loc, np.float32, order="C")
    des = np.array(des, np.uint8, order="C")
    fn = lib().orc_sift_convert_a2b
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    did = fn(_p(loc), _p(des), len(des))
    return bool(did), loc, des


def matchtxt_record(name0, name1, matches0):
    m = np.ascontiguousarray(matches0, np.int32)
    buf = C.create_string_buffer(64 + 32 * max(len(m), 1))
    ln = lib().orc_matchtxt_record(os.fsencode(name0), os.fsencode(name1), _p(m), len(m), buf, len(buf))
    return buf.raw[:ln].decode()


def image_pairs(n, consecutive, discrete):
    disc = np.ascontiguousarray(discrete, np.int32)
    cap = n * (consecutive + len(disc)) + 1
    out = np.empty((cap, 2), np.int32)
    c = lib().orc_image_pairs(n, consecutive, _p(disc), len(disc), _p(out), cap)
    return out[:c]


def mat_write_pmatch(image1, image2, fc1, fc2, idx1, idx2, reservation=0):
    a = np.ascontiguousarray(idx1, np.int32)
    b = np.ascontiguousarray(idx2, np.int32)
    return lib().orc_mat_write_pmatch(os.fsencode(image1), os.fsencode(image2), fc1, fc2, len(a), _p(a), _p(b), reservation)


def mat_write_fmatch(image1, image2, idx1, idx2, F):
    a = np.ascontiguousarray(idx1, np.int32)
    b = np.ascontiguousarray(idx2, np.int32)
    Fm = np.ascontiguousarray(F, np.float32).reshape(9)
    return lib().orc_mat_write_fmatch(os.fsencode(image1), os.fsencode(image2), len(a), _p(a), _p(b), _p(Fm))


def mat_get_pmatch(image, other, fc_other, cap=1 << 20):
    a = np.empty(cap, np.int32)
    b = np.empty(cap, np.int32)
    nm = lib().orc_mat_get_pmatch(os.fsencode(image), os.fsencode(other), fc_other, _p(a), _p(b), cap)
    return a[:nm], b[:nm]


def geom_pair(xy0, xy1, idx0, idx1, max_sampson=2.0, hypotheses=256, seed=0, pair_pos=0, min_inliers=0):
    """The library's stated RANSAC / eight-point / Sampson procedure for one pair, in scalar C.
    Returns (F [3, 3] float32, inlier idx0, inlier idx1)."""
    xy0 = np.ascontiguousarray(xy0, np.float32)
    xy1 = np.ascontiguousarray(xy1, np.float32)
    idx0 = np.ascontiguousarray(idx0, np.int32)
    idx1 = np.ascontiguousarray(idx1, np.int32)
    n = len(idx0)
    F = np.zeros(9, np.float32)
    in0 = np.empty(max(n, 1), np.int32)
    in1 = np.empty(max(n, 1), np.int32)
    fn = lib().oracle_geom_pair
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int,
                   C.c_uint, C.c_ulonglong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    nf = fn(_p(xy0), len(xy0), _p(xy1), len(xy1), _p(idx0), _p(idx1), n, float(max_sampson), int(hypotheses),
            int(seed) & 0xFFFFFFFF, int(pair_pos), int(min_inliers), _p(F), _p(in0), _p(in1))
    return F.reshape(3, 3), in0[:nf], in1[:nf]


def pose_pair(xy0, xy1, idx0, idx1, F, cam):
    """The library's stated pose recovery for one pair, in scalar C: (R [3, 3], T [3], NE)."""
    xy0 = np.ascontiguousarray(xy0, np.float32)
    xy1 = np.ascontiguousarray(xy1, np.float32)
    idx0 = np.ascontiguousarray(idx0, np.int32)
    idx1 = np.ascontiguousarray(idx1, np.int32)
    F = np.ascontiguousarray(F, np.float32).reshape(9)
    cam = np.ascontiguousarray(cam, np.float32).reshape(8)
    R, T = np.zeros(9, np.float32), np.zeros(3, np.float32)
    fn = lib().oracle_pose_pair
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p]
    ne = fn(_p(xy0), len(xy0), _p(xy1), len(xy1), _p(idx0), _p(idx1), len(idx0), _p(F), _p(cam), _p(R), _p(T))
    return R.reshape(3, 3), T, ne


def shortlist(words_per_image, nwords, topk):
    """Pair shortlist from per-image word assignments; returns (pairs [n, 2], scores [nimg, nimg])."""
    nimg = len(words_per_image)
    off = np.zeros(nimg + 1, np.int64)
    for i, w in enumerate(words_per_image):
        off[i + 1] = off[i] + len(w)
    words = np.ascontiguousarray(np.concatenate([np.asarray(w, np.int32) for w in words_per_image] + [np.zeros(0, np.int32)]))
    cap = max(nimg * (nimg - 1) // 2, 1)
    pairs = np.empty((cap, 2), np.int32)
    scores = np.zeros((max(nimg, 1), max(nimg, 1)), np.int32)
    fn = lib().orc_shortlist
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
    n = fn(_p(words), _p(off), nimg, int(nwords), int(topk), _p(pairs), cap, _p(scores))
    return pairs[:n], scores[:nimg, :nimg]


def train_vocabulary(images, nwords, iterations, threads=THREADS):
    rows = np.ascontiguousarray(np.concatenate([np.asarray(i, np.uint8).reshape(-1, 128) for i in images]))
    words = np.empty((max(int(nwords), 1), 128), np.uint8)
    fn = lib().orc_train_vocabulary
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
    k = fn(_p(rows), len(rows), int(nwords), int(iterations), threads, _p(words))
    return words[:k]
