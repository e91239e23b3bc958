"""Host-side mirror of the reference's Python interface around the matcher.

Function names, argument meaning and file output follow
``Others/SuperPoint_data/main.py`` (the exporter that writes ``matching.txt`` and
``sift/<i>.txt`` for VisualSfM); the heavy lifting is in libvsm.so.
"""
import ctypes as C
import os

import numpy as np

from . import _binding as B


def _b(s):
    return os.fsencode(s)


# ---- pair lists (main.py:157-169) -----------------------------------------------------------
def generate_imagePairs(list_img, consecutive=3, discrete=(7, 15)):
    """Pairs (i, i+1..i+consecutive) and (i, i+d) for d in discrete, as [name_i, name_j] lists.

    ``list_img`` plays the role of column 1 of the reference's infor_file.txt."""
    n = len(list_img)
    disc = np.ascontiguousarray(list(discrete), dtype=np.int32)
    cap = n * (consecutive + len(disc)) + 1
    out = np.empty((cap, 2), np.int32)
    cnt = B.lib().vsm_image_pairs(n, int(consecutive), B._ptr(disc), len(disc), B._ptr(out), cap)
    if cnt < 0:
        raise B.VsmError(cnt, "vsm_image_pairs")
    return [[list_img[a], list_img[b]] for a, b in out[:cnt]]


def generate_exhaustive_pairs(n):
    """All i < j index pairs, int32 [n(n-1)/2, 2]."""
    cap = n * (n - 1) // 2
    out = np.empty((max(cap, 1), 2), np.int32)
    cnt = B.lib().vsm_image_pairs_exhaustive(int(n), B._ptr(out), cap)
    if cnt < 0:
        raise B.VsmError(cnt, "vsm_image_pairs_exhaustive")
    return out[:cnt]


# ---- re-sort of a pair by detector score (main.py:75-112, used at main.py:253-266) --------------
def sort_scores(scores):
    """Scores in descending order and the original index of each (main.py:75-83): the flipped ascending
    sort, so equal scores come out in the reverse of numpy's argsort order, as in the reference."""
    scores = np.asarray(scores)
    return np.flip(np.sort(scores)), np.flip(np.argsort(scores))


def sort_feature_Or_descriptors(data, sorted_index):
    """Rows (or entries of a 1-D array) of `data` in the order of `sorted_index`, as float64 like the
    reference's zero-filled result (main.py:85-102); one gather instead of a Python loop."""
    data = np.asarray(data)
    sorted_index = np.asarray(sorted_index)
    if len(data) != len(sorted_index):
        raise AssertionError("data and sorted_index differ in length")
    return data[sorted_index].astype(np.float64)


def get_correct_new_matching_index(old_match0, new_index1):
    """Dense matches (-1 = none) that name image-1 keypoints by their OLD index, re-expressed in the new order
    `new_index1` (new position p holds old keypoint new_index1[p]) (main.py:104-112).  The reference searches
    new_index1 once per match; this inverts the permutation once."""
    old = np.asarray(old_match0)
    new_index1 = np.asarray(new_index1).astype(np.int64)
    inv = np.empty(len(new_index1), np.int64)
    inv[new_index1] = np.arange(len(new_index1))
    out = -np.ones(len(old))
    has = old != -1
    out[has] = inv[old[has].astype(np.int64)]
    return out


def resort_pair_by_score(kpts0, scores0, kpts1, scores1, matches0):
    """The exporter's re-sort of one pair (main.py:253-266): both images' keypoints by descending score and
    the dense matches of image 0 carried along and re-indexed.  Returns (n_kpts0, n_kpts1, n_matches)."""
    _, i0 = sort_scores(scores0)
    _, i1 = sort_scores(scores1)
    n_matches = sort_feature_Or_descriptors(matches0, i0)
    return (sort_feature_Or_descriptors(kpts0, i0), sort_feature_Or_descriptors(kpts1, i1),
            get_correct_new_matching_index(n_matches, i1))


# ---- match txt (main.py:114-155) -------------------------------------------------------------
def get_list_matching_pairs(matches):
    """Dense matches (-1 = none) -> (list_0, list_1, number_match) strings as the reference builds them."""
    matches = np.asarray(matches)
    length = len(matches)
    idx0 = np.nonzero(matches != -1)[0]
    parts0, parts1 = [], []
    for i in idx0:
        sep = "" if i == length - 1 else " "
        parts0.append(str(int(i)) + sep)
        parts1.append(str(int(matches[i])) + sep)
    return "".join(parts0), "".join(parts1), len(idx0)


def save_matching_sfm(path, matches0, imgPath0, imgPath1):
    """Appends one ``name0 name1 N / idx0... / idx1...`` record; returns N."""
    m = np.ascontiguousarray(matches0, dtype=np.int32)
    rc = B.lib().vsm_matchtxt_append_dense(_b(path), _b(imgPath0), _b(imgPath1), B._ptr(m), len(m))
    if rc < 0:
        raise B.VsmError(rc, "vsm_matchtxt_append_dense(%s)" % path)
    return rc


def save_matching_list(path, idx0, idx1, n0, imgPath0, imgPath1):
    i0 = np.ascontiguousarray(idx0, dtype=np.int32)
    i1 = np.ascontiguousarray(idx1, dtype=np.int32)
    rc = B.lib().vsm_matchtxt_append(_b(path), _b(imgPath0), _b(imgPath1), B._ptr(i0), B._ptr(i1), len(i0), int(n0))
    if rc < 0:
        raise B.VsmError(rc, "vsm_matchtxt_append(%s)" % path)
    return rc


# ---- keypoint txt (main.py:133-140) ------------------------------------------------------------
def save_feature_locations(feature_points, path):
    xy = np.ascontiguousarray(np.asarray(feature_points)[:, :2], dtype=np.float32)
    rc = B.lib().vsm_feature_txt_write(_b(path), B._ptr(xy), len(xy))
    if rc < 0:
        raise B.VsmError(rc, "vsm_feature_txt_write(%s)" % path)


def read_feature_locations(path):
    n = B.lib().vsm_feature_txt_read(_b(path), None, 0)
    if n < 0:
        raise B.VsmError(n, "vsm_feature_txt_read(%s)" % path)
    xy = np.zeros((max(n, 1), 2), np.float32)
    B.lib().vsm_feature_txt_read(_b(path), B._ptr(xy), n)
    return xy[:n]


# ---- the converter program (test_main.cpp:30-60) ------------------------------------------------------
def sift2txt(sift_path, txt_path, what="both"):
    """FeatureData::ReadSIFTB + Sift2Txt (or write_out_data for "locations" / "descriptors")."""
    code = {"both": 0, "locations": 1, "descriptors": 2}[what]
    n = B.lib().vsm_sift_to_txt(_b(sift_path), _b(txt_path), code)
    if n < 0:
        raise B.VsmError(n, "vsm_sift_to_txt(%s)" % sift_path)
    return n


def txt2sift(txt_path, sift_path):
    """FeatureData::ReadLocFromText + saveSIFTB2: keypoint txt -> .sift with placeholder descriptors."""
    n = B.lib().vsm_txt_to_sift(_b(txt_path), _b(sift_path))
    if n < 0:
        raise B.VsmError(n, "vsm_txt_to_sift(%s)" % txt_path)
    return n


# ---- .sift (FeaturePoints.cpp:228-280, 388-446) ---------------------------------------------------
def read_sift(path, nmax=None):
    """Returns (loc float32 [n, nloc_dim], des uint8 [n, 128]) or raises."""
    npoint, nloc = C.c_int(0), C.c_int(0)
    ok = B.lib().vsm_sift_read(_b(path), None, None, 0, C.byref(npoint), C.byref(nloc))
    if not ok:
        raise B.VsmError(-6, "cannot read SIFT file %s" % path)
    n = npoint.value if nmax is None else min(nmax, npoint.value)
    loc = np.empty((n, nloc.value), np.float32)
    des = np.empty((n, 128), np.uint8)
    # with buffers the call returns the rows it put there, min(nmax, rows in the file), as the reference's does
    if n > 0 and B.lib().vsm_sift_read(_b(path), B._ptr(loc), B._ptr(des), n, None, None) != n:
        raise B.VsmError(-6, "short SIFT file %s" % path)
    return loc, des


def write_sift(path, loc, des):
    loc = np.ascontiguousarray(loc, dtype=np.float32)
    des = np.ascontiguousarray(des, dtype=np.uint8)
    if des.ndim != 2 or des.shape[1] != 128 or loc.shape[0] != des.shape[0]:
        raise ValueError("expected loc [n, d] and des [n, 128]")
    if not B.lib().vsm_sift_write(_b(path), B._ptr(loc), loc.shape[1], B._ptr(des), des.shape[0]):
        raise B.VsmError(-5, "cannot write %s" % path)


# ---- ASCII .sift, sorting, the A2B conversion (FeaturePoints.cpp:54-125, 205-226, 655-765) ---------
def read_sift_ascii(path, nmax=None, sort=False):
    """FeatureData::ReadSIFTA: (loc float32 [n, 5] = (x + 0.5, y + 0.5, 0, scale, -orientation), des uint8 [n, 128]).
    sort=True does what ReadSIFTA_DES does (FeaturePoints.cpp:498-513): SortSIFT when the first feature's
    scale is below the last one's, before the first nmax are taken."""
    npoint = C.c_int(0)
    if not B.lib().vsm_sift_read_ascii(_b(path), None, None, 0, C.byref(npoint)):
        raise B.VsmError(-6, "cannot read ASCII SIFT file %s" % path)
    total = npoint.value
    n = total if (nmax is None or sort) else min(nmax, total)
    loc = np.empty((n, 5), np.float32)
    des = np.empty((n, 128), np.uint8)
    if not B.lib().vsm_sift_read_ascii(_b(path), B._ptr(loc), B._ptr(des), n, None):
        raise B.VsmError(-6, "malformed ASCII SIFT file %s" % path)
    if sort and n > 0 and loc[0, 3] < loc[n - 1, 3]:
        loc, des, _ = sort_sift(loc, des)
    if nmax is not None:
        loc, des = loc[:nmax], des[:nmax]
    return loc, des


def write_sift_ascii(path, loc, des):
    """FeatureData::saveSIFTA for loc [n, 5]."""
    loc = np.ascontiguousarray(loc, dtype=np.float32)
    des = np.ascontiguousarray(des, dtype=np.uint8)
    if des.ndim != 2 or des.shape[1] != 128 or loc.shape != (des.shape[0], 5):
        raise ValueError("expected loc [n, 5] and des [n, 128]")
    rc = B.lib().vsm_sift_write_ascii(_b(path), B._ptr(loc), B._ptr(des), des.shape[0])
    if rc < 0:
        raise B.VsmError(rc, "vsm_sift_write_ascii(%s)" % path)
    return rc


def sort_sift(loc, des):
    """FeatureData::SortSIFT on copies: (loc, des, order) by descending scale; order[i] = old row of new row i."""
    loc = np.array(loc, dtype=np.float32, order="C")
    des = np.array(des, dtype=np.uint8, order="C")
    order = np.empty(max(len(des), 1), np.int32)
    rc = B.lib().vsm_sift_sort_by_scale(B._ptr(loc), loc.shape[1], B._ptr(des), len(des), B._ptr(order))
    if rc < 0:
        raise B.VsmError(rc, "vsm_sift_sort_by_scale")
    return loc, des, order[:len(des)]


def renormalise_sift(loc, des):
    """The in-memory part of FeatureData::ConvertA2B on copies: (converted?, loc, des)."""
    loc = np.array(loc, dtype=np.float32, order="C")
    des = np.array(des, dtype=np.uint8, order="C")
    rc = B.lib().vsm_sift_renormalise(B._ptr(loc), loc.shape[1], B._ptr(des), len(des))
    if rc < 0:
        raise B.VsmError(rc, "vsm_sift_renormalise")
    return bool(rc), loc, des


def convert_a2b(path):
    """ReadSIFTA + FeatureData::ConvertA2B: the binary .sift takes the ASCII file's place; returns (name written
    or None, feature count)."""
    out = C.create_string_buffer(4096)
    n = B.lib().vsm_sift_ascii_to_binary(_b(path), out, len(out))
    if n < 0:
        raise B.VsmError(n, "vsm_sift_ascii_to_binary(%s)" % path)
    return (os.fsdecode(out.value) if out.value else None), n


# ---- .mat (MatchFile.cpp:751-958) -------------------------------------------------------------------
def write_pmatch(image1, image2, fc1, fc2, idx1, idx2, reservation=0):
    i1 = np.ascontiguousarray(idx1, dtype=np.int32)
    i2 = np.ascontiguousarray(idx2, dtype=np.int32)
    rc = B.lib().vsm_mat_write_pmatch(_b(image1), _b(image2), int(fc1), int(fc2), len(i1), B._ptr(i1), B._ptr(i2),
                                      int(reservation))
    if rc < 0:
        raise B.VsmError(rc, "vsm_mat_write_pmatch(%s, %s)" % (image1, image2))


class MatWriter:
    """Many write_pmatch calls with every file's directory kept in memory (written back on close)."""

    def __init__(self, reservation=0, max_open=0):
        self._w = B.lib().vsm_mat_writer_open(int(reservation), int(max_open))

    def put(self, image1, image2, fc1, fc2, idx1, idx2):
        i1 = np.ascontiguousarray(idx1, dtype=np.int32)
        i2 = np.ascontiguousarray(idx2, dtype=np.int32)
        rc = B.lib().vsm_mat_writer_put(self._w, _b(image1), _b(image2), int(fc1), int(fc2), len(i1), B._ptr(i1), B._ptr(i2))
        if rc < 0:
            raise B.VsmError(rc, "vsm_mat_writer_put(%s, %s)" % (image1, image2))

    def put_fmatch(self, image1, image2, idx1, idx2, F):
        """Geometry and inlier lists of a pair whose putative record was put before."""
        idx1 = np.ascontiguousarray(idx1, np.int32)
        idx2 = np.ascontiguousarray(idx2, np.int32)
        F = np.ascontiguousarray(F, np.float32).reshape(9)
        rc = B.lib().vsm_mat_writer_put_fmatch(self._w, _b(image1), _b(image2), len(idx1), B._ptr(idx1), B._ptr(idx2), B._ptr(F))
        if rc < 0:
            raise IOError("vsm_mat_writer_put_fmatch failed (%d)" % rc)
        return rc

    def close(self):
        if self._w:
            rc = B.lib().vsm_mat_writer_close(self._w)
            self._w = None
            if rc < 0:
                raise B.VsmError(rc, "vsm_mat_writer_close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def get_pmatch(image, other, fc_other):
    nm = B.lib().vsm_mat_get_pmatch(_b(image), _b(other), int(fc_other), None, None, 0)
    a = np.empty(max(nm, 1), np.int32)
    b = np.empty(max(nm, 1), np.int32)
    if nm > 0:
        B.lib().vsm_mat_get_pmatch(_b(image), _b(other), int(fc_other), B._ptr(a), B._ptr(b), nm)
    return a[:nm], b[:nm]


def get_pmatch_r(image, other, fc_other):
    """GetPMatchR (MatchFile.cpp:897-936): (indices in `other`, indices in `image`)."""
    a, b = get_pmatch(image, other, fc_other)
    return b, a


def get_match_count(image, other):
    nm, nf = C.c_int(0), C.c_int(0)
    ok = B.lib().vsm_mat_get_match_count(_b(image), _b(other), C.byref(nm), C.byref(nf))
    return (nm.value, nf.value) if ok else None


# ---- float descriptors ---------------------------------------------------------------------------
def nearest_embed(matcher, x, emb):
    """Forward of the reference's ``nearest_embed`` (Graph_Idea/model/model_2_0.py:43-78, 102-103):
    x (batch, emb_dim, *), emb (emb_dim, num_emb) -> (quantised x, argmin)."""
    return matcher.nearest_embed(x, emb, want_result=True)


def mutual_nn_match(matcher, desc0, desc1, match_threshold=0.2, descriptor_dim=None):
    """The mutual-argmax block of the reference matcher (superglue.py:267-293) applied to the
    raw score matrix desc0^T desc1 / sqrt(d).  desc0 [d, n], desc1 [d, m] (the reference's
    native layout).  Returns the same four arrays as the reference's dict."""
    desc0 = np.ascontiguousarray(desc0, np.float32)
    desc1 = np.ascontiguousarray(desc1, np.float32)
    d = desc0.shape[0] if descriptor_dim is None else descriptor_dim
    scale = 1.0 / float(d) ** 0.5
    s0 = matcher.upload_f32(desc0, B.LAYOUT_DIM_MAJOR)
    s1 = matcher.upload_f32(desc1, B.LAYOUT_DIM_MAJOR)
    try:
        i0, d0, _ = matcher.knn2(s0, s1, B.METRIC_DOT)
        i1, d1, _ = matcher.knn2(s1, s0, B.METRIC_DOT)
    finally:
        matcher.release(s0)
        matcher.release(s1)
    n, m = desc0.shape[1], desc1.shape[1]
    mutual0 = np.arange(n) == i1[i0]
    mutual1 = np.arange(m) == i0[i1]
    mscores0 = np.where(mutual0, np.exp(d0 * np.float32(scale)), np.float32(0)).astype(np.float32)
    mscores1 = np.where(mutual1, mscores0[i1], np.float32(0)).astype(np.float32)
    valid0 = mutual0 & (mscores0 > match_threshold)
    valid1 = mutual1 & valid0[i1]
    return {
        "matches0": np.where(valid0, i0, -1).astype(np.int32),
        "matches1": np.where(valid1, i1, -1).astype(np.int32),
        "matching_scores0": mscores0,
        "matching_scores1": mscores1,
    }


def write_fmatch(image1, image2, idx1, idx2, F):
    """MatchFile::WriteFMatch (MatchFile.cpp:596-605): geometry and inlier lists into both images' records."""
    idx1 = np.ascontiguousarray(idx1, np.int32)
    idx2 = np.ascontiguousarray(idx2, np.int32)
    F = np.ascontiguousarray(F, np.float32).reshape(9)
    rc = B.lib().vsm_mat_write_fmatch(_b(image1), _b(image2), len(idx1), B._ptr(idx1), B._ptr(idx2), B._ptr(F))
    if rc < 0:
        raise IOError("vsm_mat_write_fmatch failed (%d)" % rc)
    return rc


def write_ematch(image1, image2, idx1, idx2, F, ne, R, T, f1, f2):
    """MatchFile::WriteIMatch with the whole TwoViewGeometry (MatchFile.cpp:643-748, MatchFile.h:68-83): F and
    the inlier lists as write_fmatch, plus NE, R, T and the focal lengths F1, F2; the second image's file gets
    the exchanged view (SetGeometryR, MatchFile.h:121-141)."""
    idx1 = np.ascontiguousarray(idx1, np.int32)
    idx2 = np.ascontiguousarray(idx2, np.int32)
    F = np.ascontiguousarray(F, np.float32).reshape(9)
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    T = np.ascontiguousarray(T, np.float32).reshape(3)
    rc = B.lib().vsm_mat_write_ematch(_b(image1), _b(image2), len(idx1), B._ptr(idx1), B._ptr(idx2), B._ptr(F), int(ne),
                                      B._ptr(R), B._ptr(T), float(f1), float(f2))
    if rc < 0:
        raise IOError("vsm_mat_write_ematch failed (%d)" % rc)
    return rc


def get_pose(image, other, fc_other):
    """The pose fields of a match record as stored in `image`'s file: (NM, NE, R [3, 3], T [3], (F1, F2))."""
    import ctypes as C
    R, T, f12 = np.zeros((3, 3), np.float32), np.zeros(3, np.float32), np.zeros(2, np.float32)
    ne = C.c_int(0)
    nm = B.lib().vsm_mat_get_pose(_b(image), _b(other), int(fc_other), C.byref(ne), B._ptr(R), B._ptr(T), B._ptr(f12))
    return nm, ne.value, R, T, (float(f12[0]), float(f12[1]))


def estimate_pose(matcher, kpts0, kpts1, K0, K1, thresh, samples=256, seed=0):
    """The reference's estimate_pose (Others/SuperPoint_data/models/utils.py:288-312), same call and return
    shape: matched keypoint arrays and the two 3 x 3 intrinsic matrices in, (R, t, inlier mask) out, None
    with fewer than 5 matches or no pose.  `thresh` is in pixels (the reference divides it by the mean focal
    length because it works on normalised coordinates; here the model is fitted in pixel coordinates).  The
    estimator is this library's (fundamental matrix by RANSAC, then E = K1^T F K0 decomposed, the
    decomposition with most inliers in front of both cameras), not OpenCV's five-point solver."""
    kpts0 = np.ascontiguousarray(kpts0, np.float32)
    kpts1 = np.ascontiguousarray(kpts1, np.float32)
    if len(kpts0) < 5:
        return None
    got = estimate_fundamental(matcher, kpts0, kpts1, thresh, samples, seed)
    if got is None:
        return None
    F, mask = got
    sel = np.nonzero(mask)[0].astype(np.int32)
    if len(sel) == 0:
        return None
    K0, K1 = np.asarray(K0, np.float64), np.asarray(K1, np.float64)
    cams = np.asarray([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], K1[0, 0], K1[1, 1], K1[0, 2], K1[1, 2]], np.float32)
    s0, s1 = matcher.upload_xy(kpts0), matcher.upload_xy(kpts1)
    try:
        R, T, ne = matcher.recover_pose([[s0, s1]], cams, F, [0, len(sel)], sel, sel)
    finally:
        matcher.release(s0)
        matcher.release(s1)
    if ne[0] <= 0:
        return None
    return R[0], T[0], mask


def _get_imatch(fn, image, other, fc_other):
    import ctypes as C
    F = np.zeros((3, 3), np.float32)
    nf = C.c_int(0)
    cap = max(int(fc_other), 1)
    r0, r1 = np.empty(cap, np.int32), np.empty(cap, np.int32)
    nm = fn(_b(image), _b(other), int(fc_other), B._ptr(F), C.byref(nf), B._ptr(r0), B._ptr(r1), cap)
    return nm, F, r0[:nf.value], r1[:nf.value]


def get_imatch(image, other, fc_other):
    """MatchFile::GetIMatch (MatchFile.cpp:961-1021): (NM, F, inliers[0], inliers[1])."""
    return _get_imatch(B.lib().vsm_mat_get_imatch, image, other, fc_other)


def get_imatch_r(image, other, fc_other):
    """MatchFile::GetIMatchR (MatchFile.cpp:1025-1084)."""
    return _get_imatch(B.lib().vsm_mat_get_imatch_r, image, other, fc_other)


def estimate_fundamental(matcher, kpts0, kpts1, thresh, samples=256, seed=0):
    """Geometry of one pair from its matched keypoints, in the calling style of estimate_pose
    (models/utils.py:288-312: matched keypoint arrays in, model and inlier mask out; None when there
    are too few matches).  Returns (F [3, 3], mask [n] bool) with x1^T F x0 = 0; `thresh` is the Sampson
    distance in pixels.  estimate_pose goes on from here to the relative pose with the camera intrinsics."""
    kpts0 = np.ascontiguousarray(kpts0, np.float32)
    kpts1 = np.ascontiguousarray(kpts1, np.float32)
    n = len(kpts0)
    if n < 8:
        return None
    s0, s1 = matcher.upload_xy(kpts0), matcher.upload_xy(kpts1)
    try:
        idx = np.arange(n, dtype=np.int32)
        F, nf, in_off, i0, _ = matcher.verify_matches([(s0, s1)], np.array([0, n], np.int64), idx, idx,
                                                      max_sampson=thresh, hypotheses=samples, seed=seed)
    finally:
        matcher.release(s0)
        matcher.release(s1)
    if nf[0] == 0:
        return None
    mask = np.zeros(n, bool)
    mask[i0] = True
    return F[0], mask
