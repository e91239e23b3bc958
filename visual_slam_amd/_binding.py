"""ctypes binding of libvsm.so (C ABI: include/vsm.h)."""
import ctypes as C
import os
from fractions import Fraction

import numpy as np

METRIC_DOT, METRIC_L2 = 0, 1
LAYOUT_ROWS, LAYOUT_DIM_MAJOR = 0, 1
NO_SECOND = 2**31 - 1
F32_FAST_EXACT = 0
F32_MFMA_F32 = 1
F32_REVERSE_AUTO, F32_REVERSE_CANDIDATES, F32_REVERSE_ALL = 0, 1, 2
U8_AUTO = 0
U8_PACKED = 1
U8_TILE_SEPARATE = 2
U8_TILE_FUSED = 3

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    return os.path.join(_HERE, "libvsm.so")


class VsmError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("vsm error %d: %s" % (code, message))
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("ratio_num", C.c_int32),
        ("ratio_den", C.c_int32),
        ("mutual", C.c_int32),
        ("reserved", C.c_int32),
        ("max_dist_sq", C.c_int64),
        ("min_score", C.c_float),
        ("max_dist_sq_f", C.c_float),
    ]


_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); one entry per declaration in include/vsm.h
SIGNATURES = {
    "vsm_create": (C.c_void_p, [C.c_int]),
    "vsm_destroy": (None, [C.c_void_p]),
    "vsm_last_error": (C.c_char_p, [C.c_void_p]),
    "vsm_synchronize": (C.c_int, [C.c_void_p]),
    "vsm_stream": (C.c_void_p, [C.c_void_p]),
    "vsm_upload_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_upload_u8_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_upload_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vsm_upload_f32_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vsm_set_scratch_budget": (C.c_int, [C.c_void_p, C.c_longlong]),
    "vsm_set_f32_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "vsm_set_f32_reverse": (C.c_int, [C.c_void_p, C.c_int]),
    "vsm_last_f32_reverse": (C.c_int, [C.c_void_p]),
    "vsm_last_f32_path": (C.c_int, [C.c_void_p]),
    "vsm_set_u8_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "vsm_last_u8_path": (C.c_int, [C.c_void_p]),
    "vsm_last_u8_rescanned": (C.c_int, [C.c_void_p]),
    "vsm_release_set": (C.c_int, [C.c_void_p, C.c_int]),
    "vsm_set_count": (C.c_int, [C.c_void_p, C.c_int]),
    "vsm_match_pair": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_match_pair_noalloc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int]),
    "vsm_match_pair_dense": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "vsm_match_pairs": (C.c_longlong, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Params), C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]),
    "vsm_submit_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Params), C.c_int]),
    "vsm_collect_matches": (C.c_longlong, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]),
    "vsm_match_pairs_device": (C.c_longlong, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Params), C.c_int]),
    "vsm_fetch_matches": (C.c_longlong, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]),
    "vsm_knn2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsm_knn2_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vsm_merge_top2_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsm_merge_top2_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsm_dbq_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "vsm_dbq_destroy": (None, [C.c_void_p]),
    "vsm_dbq_last_error": (C.c_char_p, [C.c_void_p]),
    "vsm_dbq_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int]),
    "vsm_dbq_slices": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vsm_dbq_query": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "vsm_nearest_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p]),
    "vsm_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "vsm_last_kernel_ms": (C.c_float, [C.c_void_p]),
    "vsm_last_verify_ms": (C.c_float, [C.c_void_p]),
    "vsm_last_submission_pairs": (C.c_int, [C.c_void_p]),
    "vsm_sift_read": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vsm_sift_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "vsm_sift_read_ascii": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "vsm_sift_write_ascii": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_sift_sort_by_scale": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vsm_sift_renormalise": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "vsm_sift_ascii_to_binary": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "vsm_feature_txt_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int]),
    "vsm_feature_txt_read": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int]),
    "vsm_matchtxt_append_dense": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]),
    "vsm_matchtxt_append": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "vsm_matchtxt_open": (C.c_void_p, [C.c_char_p, C.c_int]),
    "vsm_matchtxt_write": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "vsm_matchtxt_close": (C.c_int, [C.c_void_p]),
    "vsm_image_pairs": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "vsm_sift_to_txt": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "vsm_txt_to_sift": (C.c_int, [C.c_char_p, C.c_char_p]),
    "vsm_image_pairs_exhaustive": (C.c_longlong, [C.c_int, C.c_void_p, C.c_longlong]),
    "vsm_sample_rows_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "vsm_train_vocabulary": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vsm_shortlist_pairs": (C.c_longlong, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong,
                                           C.c_void_p]),
    "vsm_mat_write_pmatch": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_mat_set_locking": (C.c_int, [C.c_int]),
    "vsm_mat_writer_open": (C.c_void_p, [C.c_int, C.c_int]),
    "vsm_mat_writer_put": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vsm_mat_writer_close": (C.c_int, [C.c_void_p]),
    "vsm_mat_get_pmatch": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_mat_get_pmatch_r": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_mat_get_match_count": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vsm_upload_xy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_verify_matches": (C.c_longlong, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_longlong]),
    "vsm_recover_pose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsm_mat_write_ematch": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_float]),
    "vsm_mat_writer_put_ematch": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float]),
    "vsm_mat_get_pose": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsm_mat_write_fmatch": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsm_mat_writer_put_fmatch": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsm_mat_get_imatch": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                     C.c_void_p, C.c_void_p, C.c_int]),
    "vsm_mat_get_imatch_r": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                       C.c_void_p, C.c_void_p, C.c_int]),
}


class GeomParams(C.Structure):
    """vsm_geom_params"""
    _fields_ = [("max_sampson", C.c_float), ("hypotheses", C.c_int), ("seed", C.c_uint), ("min_inliers", C.c_int),
                ("first_pair", C.c_longlong)]

_lib = None


def lib():
    """Loads libvsm.so once.  Raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise VsmError(-2, "%s is missing: build it with `make -C visual_slam_amd/csrc`" % path)
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def ratio_to_fraction(ratio):
    """Lowe ratio as an exact rational (the device test is integer-exact)."""
    if ratio is None or ratio <= 0:
        return 0, 1
    fr = Fraction(float(ratio)).limit_denominator(1000)
    return fr.numerator, fr.denominator


def make_params(ratio=0.8, mutual=True, max_dist_sq=-1, min_score=float("-inf"), max_dist_sq_f=-1.0):
    num, den = ratio_to_fraction(ratio)
    return Params(num, den, 1 if mutual else 0, 0, int(max_dist_sq), float(min_score), float(max_dist_sq_f))


class Matcher:
    """One matching context on one GPU (single-threaded, like the reference's classes)."""

    def __init__(self, device=0):
        self._l = lib()
        self._h = self._l.vsm_create(int(device))
        if not self._h:
            raise VsmError(-2, self._l.vsm_last_error(None).decode())
        self._kinds = {}
        self._inflight = []

    def close(self):
        if getattr(self, "_h", None):
            self._l.vsm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc < 0:
            raise VsmError(rc, self._l.vsm_last_error(self._h).decode())
        return rc

    # ---- sets -----------------------------------------------------------------------
    def upload_u8(self, des):
        des = np.ascontiguousarray(des, dtype=np.uint8)
        if des.ndim != 2 or des.shape[1] != 128:
            raise ValueError("expected uint8 [n, 128]")
        sid = self._check(self._l.vsm_upload_u8(self._h, _ptr(des), des.shape[0]))
        self._kinds[sid] = ("u8", des.shape[0])
        return sid

    def upload_u8_device(self, dev_ptr, n):
        sid = self._check(self._l.vsm_upload_u8_device(self._h, C.c_void_p(int(dev_ptr)), int(n)))
        self._kinds[sid] = ("u8", int(n))
        return sid

    def upload_f32(self, des, layout=LAYOUT_ROWS):
        des = np.ascontiguousarray(des, dtype=np.float32)
        if des.ndim != 2:
            raise ValueError("expected float32 2-D array")
        n, dim = (des.shape[0], des.shape[1]) if layout == LAYOUT_ROWS else (des.shape[1], des.shape[0])
        sid = self._check(self._l.vsm_upload_f32(self._h, _ptr(des), n, dim, layout))
        self._kinds[sid] = ("f32", n)
        return sid

    def upload_f32_device(self, dev_ptr, n, dim, layout=LAYOUT_ROWS):
        sid = self._check(self._l.vsm_upload_f32_device(self._h, C.c_void_p(int(dev_ptr)), int(n), int(dim), layout))
        self._kinds[sid] = ("f32", int(n))
        return sid

    def release(self, sid):
        self._check(self._l.vsm_release_set(self._h, sid))
        self._kinds.pop(sid, None)

    def count(self, sid):
        return self._check(self._l.vsm_set_count(self._h, sid))

    # ---- matching ---------------------------------------------------------------------
    def match_pair(self, set0, set1, ratio=0.8, mutual=True, max_dist_sq=-1, metric=METRIC_L2,
                   min_score=float("-inf"), max_dist_sq_f=-1.0, return_dist=False):
        p = make_params(ratio, mutual, max_dist_sq, min_score, max_dist_sq_f)
        kind, n0 = self._kinds[set0]
        cap = max(n0, 1)
        idx0 = np.empty(cap, np.int32)
        idx1 = np.empty(cap, np.int32)
        dist = np.empty(cap, np.int32 if kind == "u8" else np.float32) if return_dist else None
        nm = self._check(self._l.vsm_match_pair(self._h, set0, set1, C.byref(p), metric,
                                                _ptr(idx0), _ptr(idx1), _ptr(dist), cap))
        if return_dist:
            return idx0[:nm].copy(), idx1[:nm].copy(), dist[:nm].copy()
        return idx0[:nm].copy(), idx1[:nm].copy()

    def match_pair_noalloc(self, set0, set1, idx0_dev, idx1_dev, count_dev, cap, ratio=0.8, mutual=True, max_dist_sq=-1):
        """match_pair with all intermediates in the library's fixed per-GPU array and the lists written to
        the caller's device buffers (raw device addresses); queued on the context's stream, not waited for."""
        p = make_params(ratio, mutual, max_dist_sq)
        self._check(self._l.vsm_match_pair_noalloc(self._h, set0, set1, C.byref(p), C.c_void_p(int(idx0_dev)),
                                                   C.c_void_p(int(idx1_dev)), C.c_void_p(int(count_dev)), int(cap)))

    def match_pair_dense(self, set0, set1, ratio=0.8, mutual=True, max_dist_sq=-1, metric=METRIC_L2,
                         min_score=float("-inf"), max_dist_sq_f=-1.0):
        p = make_params(ratio, mutual, max_dist_sq, min_score, max_dist_sq_f)
        n0, n1 = self._kinds[set0][1], self._kinds[set1][1]
        m0 = np.empty(max(n0, 1), np.int32)
        m1 = np.empty(max(n1, 1), np.int32)
        self._check(self._l.vsm_match_pair_dense(self._h, set0, set1, C.byref(p), metric, _ptr(m0), _ptr(m1)))
        return m0[:n0].copy(), m1[:n1].copy()

    def match_pairs(self, pairs, ratio=0.8, mutual=True, max_dist_sq=-1, cap=None, metric=METRIC_L2,
                    min_score=float("-inf"), max_dist_sq_f=-1.0):
        """Many pairs per submission; u8 pairs or float pairs (one padded dimension per call)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        p = make_params(ratio, mutual, max_dist_sq, min_score, max_dist_sq_f)
        if cap is None:
            cap = int(sum(self._kinds[int(a)][1] for a in pairs[:, 0])) or 1
        off = np.empty(len(pairs) + 1, np.int64)
        idx0 = np.empty(cap, np.int32)
        idx1 = np.empty(cap, np.int32)
        tot = self._check(self._l.vsm_match_pairs(self._h, _ptr(pairs), len(pairs), C.byref(p), metric,
                                                  _ptr(off), _ptr(idx0), _ptr(idx1), cap))
        return off, idx0[:tot].copy(), idx1[:tot].copy()

    def submit_pairs(self, pairs, ratio=0.8, mutual=True, max_dist_sq=-1, metric=METRIC_L2,
                     min_score=float("-inf"), max_dist_sq_f=-1.0):
        """Queues one submission (at most two in flight); collect_matches returns the oldest one's lists."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        p = make_params(ratio, mutual, max_dist_sq, min_score, max_dist_sq_f)
        self._check(self._l.vsm_submit_pairs(self._h, _ptr(pairs), len(pairs), C.byref(p), metric))
        self._inflight.append((len(pairs), int(sum(self._kinds[int(a)][1] for a in pairs[:, 0])) or 1))

    def collect_matches(self, bufs=None):
        """bufs: optional (offsets, idx0, idx1) numpy arrays to reuse (large enough for the submission)."""
        npairs, cap = self._inflight.pop(0)
        if bufs is None:
            bufs = (np.empty(npairs + 1, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32))
        off, idx0, idx1 = bufs
        tot = self._check(self._l.vsm_collect_matches(self._h, _ptr(off), _ptr(idx0), _ptr(idx1), min(cap, len(idx0))))
        return off[:npairs + 1], idx0[:tot], idx1[:tot]

    def match_pairs_device(self, pairs, ratio=0.8, mutual=True, max_dist_sq=-1, metric=METRIC_L2,
                           min_score=float("-inf"), max_dist_sq_f=-1.0):
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        p = make_params(ratio, mutual, max_dist_sq, min_score, max_dist_sq_f)
        return self._check(self._l.vsm_match_pairs_device(self._h, _ptr(pairs), len(pairs), C.byref(p), metric))

    def fetch_matches(self, npairs, total):
        off = np.empty(npairs + 1, np.int64)
        idx0 = np.empty(max(total, 1), np.int32)
        idx1 = np.empty(max(total, 1), np.int32)
        self._check(self._l.vsm_fetch_matches(self._h, _ptr(off), _ptr(idx0), _ptr(idx1), max(total, 1)))
        return off, idx0[:total], idx1[:total]

    def knn2(self, set_q, set_db, metric=METRIC_L2, want_idx2=False):
        kind, nq = self._kinds[set_q]
        dt = np.int32 if kind == "u8" else np.float32
        idx1 = np.empty(max(nq, 1), np.int32)
        idx2 = np.empty(max(nq, 1), np.int32) if want_idx2 else None
        d1 = np.empty(max(nq, 1), dt)
        d2 = np.empty(max(nq, 1), dt)
        self._check(self._l.vsm_knn2(self._h, set_q, set_db, metric, _ptr(idx1), _ptr(idx2), _ptr(d1), _ptr(d2)))
        if want_idx2:
            return idx1[:nq], idx2[:nq], d1[:nq], d2[:nq]
        return idx1[:nq], d1[:nq], d2[:nq]

    def knn2_device(self, set_q, set_db, metric, packed_ptr):
        """2-NN of a float query set in one database slice, left in a device buffer of 4 * nq int32 words."""
        return self._check(self._l.vsm_knn2_device(self._h, set_q, set_db, metric, C.c_void_p(int(packed_ptr))))

    def merge_top2_device(self, gathered_ptr, base, nslices, nq, metric):
        """Device merge of an all-gathered [nslices][4][nq] buffer; returns (k1, k2, i1, i2) on the host."""
        base = np.ascontiguousarray(base, np.int64)
        ok1, ok2 = np.empty(nq, np.float32), np.empty(nq, np.float32)
        oi1, oi2 = np.empty(nq, np.int64), np.empty(nq, np.int64)
        self._check(self._l.vsm_merge_top2_device(self._h, C.c_void_p(int(gathered_ptr)), _ptr(base), nslices, nq, metric,
                                                  _ptr(ok1), _ptr(ok2), _ptr(oi1), _ptr(oi2)))
        return ok1, ok2, oi1, oi2

    def nearest_embed(self, x, emb, want_result=True):
        x = np.ascontiguousarray(x, dtype=np.float32)
        emb = np.ascontiguousarray(emb, dtype=np.float32)
        b, d = x.shape[0], x.shape[1]
        npts = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
        argmin = np.empty((b, npts), np.int64)
        res = np.empty((b, d, npts), np.float32) if want_result else None
        self._check(self._l.vsm_nearest_embed(self._h, _ptr(x), b, d, npts, _ptr(emb), emb.shape[1],
                                              _ptr(argmin), _ptr(res)))
        return (res.reshape(x.shape) if want_result else None), argmin.reshape((b,) + tuple(x.shape[2:]))

    # ---- pair list by content -----------------------------------------------------------
    def shortlist_pairs(self, sets, vocab_set, topk, want_scores=False):
        """Image pairs worth matching, from word histograms over the vocabulary ``vocab_set``: every
        image's ``topk`` best partners by weighted histogram intersection.  Returns int32 [npairs, 2]
        positions in ``sets`` (i < j, ascending), and the [n, n] score matrix if asked for."""
        sets = np.ascontiguousarray(sets, dtype=np.int32).reshape(-1)
        n = sets.shape[0]
        cap = min(n * min(int(topk), n), n * (n - 1) // 2)
        pairs = np.empty((max(cap, 1), 2), np.int32)
        scores = np.empty((n, n), np.int32) if want_scores else None
        got = self._check(self._l.vsm_shortlist_pairs(self._h, _ptr(sets), n, int(vocab_set), int(topk), _ptr(pairs),
                                                      cap, _ptr(scores)))
        return (pairs[:got], scores) if want_scores else pairs[:got]

    def sample_rows(self, sets, stride):
        """A new resident u8 set with every ``stride``-th row of the concatenation of ``sets``."""
        sets = np.ascontiguousarray(sets, dtype=np.int32).reshape(-1)
        sid = self._check(self._l.vsm_sample_rows_u8(self._h, _ptr(sets), sets.shape[0], int(stride)))
        self._kinds[sid] = ("u8", self.count(sid))
        return sid

    def train_vocabulary(self, sets, nwords, iterations=10):
        """k-means vocabulary over all rows of the resident u8 sets: uint8 [K, 128], K = min(nwords, rows)."""
        sets = np.ascontiguousarray(sets, dtype=np.int32).reshape(-1)
        words = np.empty((max(int(nwords), 1), 128), np.uint8)
        k = self._check(self._l.vsm_train_vocabulary(self._h, _ptr(sets), sets.shape[0], int(nwords), int(iterations),
                                                     _ptr(words)))
        return words[:k]

    # ---- geometric verification ---------------------------------------------------------
    def upload_xy(self, xy):
        """Keypoint locations of one image, n x (x, y)."""
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        if xy.ndim != 2 or xy.shape[1] != 2:
            raise ValueError("expected an n x 2 float32 array")
        sid = self._check(self._l.vsm_upload_xy(self._h, _ptr(xy), xy.shape[0]))
        self._kinds[sid] = ("xy", xy.shape[0])
        return sid

    def verify_matches(self, xy_pairs, offsets=None, idx0=None, idx1=None, max_sampson=2.0, hypotheses=256, seed=0,
                       min_inliers=0, cap=None, first_pair=0):
        """Fundamental matrix and inlier matches per pair.  With offsets/idx0/idx1 None the putative
        matches are the lists of the last collected submission, read on the device (cap: their total,
        as match_pairs_device returned it; without it the call may run twice to size the buffers).
        Returns (F [npairs, 3, 3], nf [npairs], in_offsets, in_idx0, in_idx1)."""
        xy_pairs = np.ascontiguousarray(xy_pairs, dtype=np.int32).reshape(-1, 2)
        npairs = xy_pairs.shape[0]
        g = GeomParams(float(max_sampson), int(hypotheses), int(seed) & 0xFFFFFFFF, int(min_inliers), int(first_pair))
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, np.int64)
            idx0 = np.ascontiguousarray(idx0, np.int32)
            idx1 = np.ascontiguousarray(idx1, np.int32)
            cap = int(offsets[-1]) if npairs else 0
        else:
            cap = int(cap) if cap is not None else 0
        F = np.zeros((npairs, 3, 3), np.float32)
        nf = np.zeros(npairs, np.int32)
        in_off = np.zeros(npairs + 1, np.int64)
        in0 = np.empty(max(cap, 1), np.int32)
        in1 = np.empty(max(cap, 1), np.int32)
        got = self._l.vsm_verify_matches(self._h, _ptr(xy_pairs), npairs, _ptr(offsets), _ptr(idx0), _ptr(idx1),
                                         C.byref(g), _ptr(F), _ptr(nf), _ptr(in_off), _ptr(in0), _ptr(in1), cap)
        if got == -4:   # VSM_ECAP: size the buffers from the offsets and run again
            cap = int(in_off[-1])
            in0 = np.empty(max(cap, 1), np.int32)
            in1 = np.empty(max(cap, 1), np.int32)
            got = self._l.vsm_verify_matches(self._h, _ptr(xy_pairs), npairs, _ptr(offsets), _ptr(idx0), _ptr(idx1),
                                             C.byref(g), _ptr(F), _ptr(nf), _ptr(in_off), _ptr(in0), _ptr(in1), cap)
        got = self._check(got)
        return F, nf, in_off, in0[:got], in1[:got]

    def recover_pose(self, xy_pairs, cameras, F, in_offsets, in_idx0, in_idx1):
        """Relative pose per pair from its fundamental matrix, the two cameras' intrinsics (cameras
        [npairs, 8] = fx, fy, cx, cy of the first image, then of the second) and its inlier matches.
        Returns (R [npairs, 3, 3], T [npairs, 3], ne [npairs]): X1 = R X0 + T, |T| = 1, ne = matches in front
        of both cameras (0: no pose)."""
        xy_pairs = np.ascontiguousarray(xy_pairs, dtype=np.int32).reshape(-1, 2)
        npairs = xy_pairs.shape[0]
        cameras = np.ascontiguousarray(cameras, np.float32).reshape(npairs, 8)
        F = np.ascontiguousarray(F, np.float32).reshape(npairs, 9)
        in_offsets = np.ascontiguousarray(in_offsets, np.int64)
        in_idx0 = np.ascontiguousarray(in_idx0, np.int32)
        in_idx1 = np.ascontiguousarray(in_idx1, np.int32)
        R = np.zeros((npairs, 3, 3), np.float32)
        T = np.zeros((npairs, 3), np.float32)
        ne = np.zeros(npairs, np.int32)
        self._check(self._l.vsm_recover_pose(self._h, _ptr(xy_pairs), npairs, _ptr(cameras), _ptr(F), _ptr(in_offsets),
                                             _ptr(in_idx0), _ptr(in_idx1), _ptr(R), _ptr(T), _ptr(ne)))
        return R, T, ne

    # ---- measurement --------------------------------------------------------------------
    def set_scratch_budget(self, nbytes):
        """Device scratch one submission of match_pairs may use (the pair list is cut accordingly)."""
        self._check(self._l.vsm_set_scratch_budget(self._h, int(nbytes)))

    def set_f32_mode(self, mode):
        """0 = bf16 coarse pass + exact re-rank (default), 1 = f32 matrix cores only; same results."""
        self._check(self._l.vsm_set_f32_mode(self._h, int(mode)))

    def set_f32_reverse(self, mode):
        """Mutual check of float pair submissions: 0 = choose by size and rule, 1 = reverse searches over the
        candidates' train rows only, 2 = both directions in one launch; same results."""
        self._check(self._l.vsm_set_f32_reverse(self._h, int(mode)))

    def last_f32_reverse(self):
        return self._l.vsm_last_f32_reverse(self._h)

    def set_u8_mode(self, mode):
        """0 = tile-minimum query pass for ratio-test submissions (default), 1 = packed form only; same results."""
        self._check(self._l.vsm_set_u8_mode(self._h, int(mode)))

    def last_u8_path(self):
        return int(self._l.vsm_last_u8_path(self._h))

    def last_u8_rescanned(self):
        """Rows of the last finished u8 submission that needed the rescan pass (0 in the packed form)."""
        return int(self._l.vsm_last_u8_rescanned(self._h))

    def last_f32_path(self):
        return int(self._l.vsm_last_f32_path(self._h))

    def set_timing(self, on=True):
        self._check(self._l.vsm_set_timing(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        return float(self._l.vsm_last_kernel_ms(self._h))

    def last_submission_pairs(self):
        return int(self._l.vsm_last_submission_pairs(self._h))

    def last_verify_ms(self):
        return float(self._l.vsm_last_verify_ms(self._h))

    def synchronize(self):
        self._check(self._l.vsm_synchronize(self._h))

    def stream(self):
        return self._l.vsm_stream(self._h)


def merge_top2_f32(k1, k2, i1, i2, base, metric):
    """Host merge of per-slice 2-NN lists (arrays [nslices, nq])."""
    k1 = np.ascontiguousarray(k1, np.float32)
    k2 = np.ascontiguousarray(k2, np.float32)
    i1 = np.ascontiguousarray(i1, np.int32)
    i2 = np.ascontiguousarray(i2, np.int32)
    base = np.ascontiguousarray(base, np.int64)
    ns, nq = k1.shape
    ok1 = np.empty(nq, np.float32)
    ok2 = np.empty(nq, np.float32)
    oi1 = np.empty(nq, np.int64)
    oi2 = np.empty(nq, np.int64)
    rc = lib().vsm_merge_top2_f32(_ptr(k1), _ptr(k2), _ptr(i1), _ptr(i2), _ptr(base), ns, nq, metric,
                                  _ptr(ok1), _ptr(ok2), _ptr(oi1), _ptr(oi2))
    if rc < 0:
        raise VsmError(rc, "vsm_merge_top2_f32")
    return ok1, ok2, oi1, oi2


class SlicedDatabase:
    """A float database cut into slices over the GPUs of this machine (vsm_dbq_*): every query's two nearest rows
    over the whole database, the per-slice lists merged after an RCCL all-gather."""

    def __init__(self, devices=(0,), nslices=None):
        self._l = lib()
        devs = np.ascontiguousarray(devices, np.int32)
        self._h = self._l.vsm_dbq_create(_ptr(devs), len(devs), int(nslices or len(devs)))
        if not self._h:
            raise VsmError(-2, self._l.vsm_dbq_last_error(None).decode())

    def _check(self, rc):
        if rc < 0:
            raise VsmError(rc, self._l.vsm_dbq_last_error(self._h).decode())
        return rc

    def load(self, rows, layout=LAYOUT_ROWS):
        rows = np.ascontiguousarray(rows, np.float32)
        n, dim = rows.shape if layout == LAYOUT_ROWS else rows.shape[::-1]
        self._check(self._l.vsm_dbq_load(self._h, _ptr(rows), n, dim, layout))
        return self

    def slices(self):
        n = self._l.vsm_dbq_slices(self._h, None)
        base = np.empty(n + 1, np.int64)
        self._l.vsm_dbq_slices(self._h, _ptr(base))
        return base

    def query(self, queries, metric=METRIC_L2, layout=LAYOUT_ROWS):
        """Returns (k1, k2, i1, i2): keys and global rows of the nearest database row and the runner-up."""
        queries = np.ascontiguousarray(queries, np.float32)
        nq = queries.shape[0] if layout == LAYOUT_ROWS else queries.shape[1]
        k1, k2 = np.empty(nq, np.float32), np.empty(nq, np.float32)
        i1, i2 = np.empty(nq, np.int64), np.empty(nq, np.int64)
        self._check(self._l.vsm_dbq_query(self._h, _ptr(queries), nq, layout, metric, _ptr(k1), _ptr(k2), _ptr(i1), _ptr(i2)))
        return k1, k2, i1, i2

    def close(self):
        if getattr(self, "_h", None):
            self._l.vsm_dbq_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
