"""Seeded random shapes and parameters through the C ABI against the CPU oracle: set sizes around the
padding and tile boundaries, ratios, mutual on/off, distance caps, float dimensions 1..256, both float
search modes, pair lists with repeated and empty sets.  Bit-exact, like the directed tests."""
import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import _binding as B
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu
EDGES = [0, 1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025]


def _size(rng):
    return int(rng.choice(EDGES)) if rng.random() < 0.5 else int(rng.integers(0, 2600))


@pytest.mark.parametrize("seed", range(64))
def test_sift_pair_random(matcher, seed):
    rng = np.random.default_rng(9000 + seed)
    n0, n1 = _size(rng), _size(rng)
    a, b = synth.sift_pair(max(n0, 1), max(n1, 1), seed=seed, planted=float(rng.uniform(0.0, 0.6)), noise=float(rng.uniform(2, 12)))
    a, b = a[:n0], b[:n1]
    if rng.random() < 0.3 and n0 and n1:       # exact duplicates: ties in both directions
        k = min(n0, n1, 20)
        b[:k] = a[rng.integers(0, n0, k)]
        a[rng.integers(0, n0, 5)] = a[0]
    num, den = (int(rng.integers(1, 12)), 12) if rng.random() < 0.8 else (0, 1)
    mutual = bool(rng.random() < 0.6)
    cap = int(rng.integers(20000, 120000)) if rng.random() < 0.3 else -1
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        g0, g1, gd = matcher.match_pair(s0, s1, ratio=(num / den if num else None), mutual=mutual, max_dist_sq=cap,
                                        return_dist=True)
        o0, o1, od = orc.match_u8(a, b, orc.params((num, den) if num else None, mutual, max_dist_sq=cap))
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
        np.testing.assert_array_equal(gd, od)
        d0, d1 = matcher.match_pair_dense(s0, s1, ratio=(num / den if num else None), mutual=mutual, max_dist_sq=cap)
        e0, e1 = orc.match_dense_u8(a, b, orc.params((num, den) if num else None, mutual, max_dist_sq=cap))
        np.testing.assert_array_equal(d0, e0)
        np.testing.assert_array_equal(d1, e1)
    finally:
        matcher.release(s0)
        matcher.release(s1)


@pytest.mark.parametrize("seed", range(40))
def test_float_pair_random(matcher, seed):
    rng = np.random.default_rng(9500 + seed)
    n0, n1 = max(_size(rng), 1), max(_size(rng), 1)
    dim = int(rng.choice([1, 2, 7, 31, 64, 65, 100, 128, 129, 200, 256]))
    a, b = synth.superpoint_pair(n0, n1, dim, seed=seed, planted=float(rng.uniform(0, 0.5)))
    if rng.random() < 0.3:
        a = (a * rng.uniform(0.2, 5.0, (n0, 1))).astype(np.float32)      # not unit length
        b = (b * rng.uniform(0.2, 5.0, (n1, 1))).astype(np.float32)
    metric = vsm.METRIC_DOT if rng.random() < 0.5 else vsm.METRIC_L2
    matcher.set_f32_mode(vsm.F32_FAST_EXACT if rng.random() < 0.7 else vsm.F32_MFMA_F32)
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    try:
        gi1, gi2, gk1, gk2 = matcher.knn2(s0, s1, metric, want_idx2=True)
        oi, ok1, ok2 = orc.knn2_f32(a, b, metric)
        np.testing.assert_array_equal(gi1, oi)
        np.testing.assert_array_equal(gk1, ok1)
        np.testing.assert_array_equal(gk2, ok2)
        ratio = (3, 4) if metric == vsm.METRIC_L2 and rng.random() < 0.5 else None
        mutual = bool(rng.random() < 0.6)
        g0, g1 = matcher.match_pair_dense(s0, s1, ratio=(0.75 if ratio else None), mutual=mutual, metric=metric, min_score=0.3)
        o0, o1, _, _ = orc.match_dense_f32(a, b, metric, orc.params(ratio, mutual, min_score=0.3))
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
    finally:
        matcher.set_f32_mode(vsm.F32_FAST_EXACT)
        matcher.release(s0)
        matcher.release(s1)


@pytest.mark.parametrize("seed", range(12))
def test_pair_list_random(matcher, seed):
    rng = np.random.default_rng(9800 + seed)
    nimg = int(rng.integers(3, 9))
    pool = synth.sift_like(400, rng)
    imgs = []
    for _ in range(nimg):
        n = _size(rng)
        d = synth.sift_like(max(n, 1), rng)[:n]
        k = min(n, 400) // 2
        if k:
            d[rng.choice(n, k, replace=False)] = np.clip(
                pool[rng.choice(400, k, replace=False)].astype(np.int32) + rng.integers(-6, 7, (k, 128)), 0, 255).astype(np.uint8)
        imgs.append(d)
    ids = [matcher.upload_u8(d) for d in imgs]
    npairs = int(rng.integers(1, 40))
    pairs = [(int(rng.integers(0, nimg)), int(rng.integers(0, nimg))) for _ in range(npairs)]   # repeats and (i, i) allowed
    mutual = bool(rng.random() < 0.5)
    if seed % 2:
        matcher.set_scratch_budget(4 << 20)      # many small submissions
    try:
        off, i0, i1 = matcher.match_pairs([[ids[a], ids[b]] for a, b in pairs], ratio=0.8, mutual=mutual)
        for k, (a, b) in enumerate(pairs):
            o0, o1, _ = orc.match_u8(imgs[a], imgs[b], orc.params((4, 5), mutual))
            np.testing.assert_array_equal(i0[off[k]:off[k + 1]], o0)
            np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o1)
    finally:
        matcher.set_scratch_budget(8 << 30)
        for s in ids:
            matcher.release(s)


@pytest.mark.parametrize("seed", range(12))
def test_geometry_random(matcher, seed):
    rng = np.random.default_rng(9900 + seed)
    npairs = int(rng.integers(1, 7))
    sets, lists = [], []
    for k in range(npairs):
        n = int(rng.choice([0, 7, 8, 9, 63, 64, 65, 200, 777, 1500]))
        xy0, xy1, i0, i1, _, _ = synth.two_view(max(n, 1), seed=100 * seed + k, inlier_fraction=float(rng.uniform(0.3, 1.0)),
                                                noise=float(rng.uniform(0.0, 1.5)))
        sets.append((xy0, xy1))
        lists.append((i0[:n], i1[:n]))
    kw = dict(max_sampson=float(rng.uniform(0.3, 4.0)), hypotheses=int(rng.integers(1, 400)), seed=int(rng.integers(0, 2**31)),
              min_inliers=int(rng.integers(0, 30)))
    first = int(rng.integers(0, 10**9))
    ids = [(matcher.upload_xy(a), matcher.upload_xy(b)) for a, b in sets]
    off = np.zeros(npairs + 1, np.int64)
    off[1:] = np.cumsum([len(l[0]) for l in lists])
    i0 = np.concatenate([l[0] for l in lists]).astype(np.int32)
    i1 = np.concatenate([l[1] for l in lists]).astype(np.int32)
    try:
        F, nf, in_off, g0, g1 = matcher.verify_matches(ids, off, i0, i1, first_pair=first, **kw)
        for p, ((a, b), (l0, l1)) in enumerate(zip(sets, lists)):
            oF, o0, o1 = orc.geom_pair(a, b, l0, l1, pair_pos=first + p, **kw)
            np.testing.assert_array_equal(F[p], oF)
            np.testing.assert_array_equal(g0[in_off[p]:in_off[p + 1]], o0)
            np.testing.assert_array_equal(g1[in_off[p]:in_off[p + 1]], o1)
    finally:
        for a, b in ids:
            matcher.release(a)
            matcher.release(b)


@pytest.mark.parametrize("seed", range(6))
def test_large_sets_random(matcher, seed):
    """Sizes that span several 8192-row chunks and 512-row strips, one call per pair and as a list."""
    rng = np.random.default_rng(9950 + seed)
    n0, n1 = int(rng.integers(7000, 21000)), int(rng.integers(7000, 21000))
    a, b = synth.sift_pair(n0, n1, seed=50 + seed, planted=0.3)
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        for mutual in (True, False):
            o0, o1, od = orc.match_u8(a, b, orc.params((4, 5), mutual))
            g0, g1, gd = matcher.match_pair(s0, s1, ratio=0.8, mutual=mutual, return_dist=True)
            np.testing.assert_array_equal(g0, o0)
            np.testing.assert_array_equal(g1, o1)
            np.testing.assert_array_equal(gd, od)
            off, i0, i1 = matcher.match_pairs([[s0, s1], [s1, s0], [s0, s1]], ratio=0.8, mutual=mutual)
            np.testing.assert_array_equal(i0[off[0]:off[1]], o0)
            np.testing.assert_array_equal(i1[off[2]:off[3]], o1)
            r0, r1, _ = orc.match_u8(b, a, orc.params((4, 5), mutual))
            np.testing.assert_array_equal(i0[off[1]:off[2]], r0)
            np.testing.assert_array_equal(i1[off[1]:off[2]], r1)
        fa, fb = synth.superpoint_pair(n0 // 3, n1 // 2, 256, seed=60 + seed)
        q0, q1 = matcher.upload_f32(fa), matcher.upload_f32(fb)
        gi1, gi2, gk1, gk2 = matcher.knn2(q0, q1, vsm.METRIC_L2, want_idx2=True)
        oi, ok1, ok2 = orc.knn2_f32(fa, fb, orc.METRIC_L2)
        np.testing.assert_array_equal(gi1, oi)
        np.testing.assert_array_equal(gk1, ok1)
        np.testing.assert_array_equal(gk2, ok2)
        matcher.release(q0)
        matcher.release(q1)
    finally:
        matcher.release(s0)
        matcher.release(s1)


@pytest.mark.parametrize("seed", range(16))
def test_shortlist_and_vocabulary_random(matcher, seed):
    """Random collections (sizes from 0, near-copies, shared rows), random vocabulary sizes, top-k and
    iteration counts: trained words, score matrix and pair list against the oracle."""
    rng = np.random.default_rng(9000 + seed)
    nimg = int(rng.integers(1, 40))
    pool = synth.sift_like(64, rng)
    images = []
    for _ in range(nimg):
        n = int(rng.choice([0, 1, 31, 200, 513, 900])) if rng.random() < 0.5 else int(rng.integers(0, 400))
        img = synth.sift_like(n, rng)
        k = int(n * rng.random() * 0.7)
        if k:
            img[:k] = np.clip(pool[rng.integers(0, 64, k)].astype(np.int32) + rng.integers(-3, 4, (k, 128)), 0, 255)
        images.append(img)
    if nimg > 2 and rng.random() < 0.5:
        images[2] = images[0].copy()
    nwords = int(rng.choice([1, 2, 33, 64, 300, 1000]))
    iterations = int(rng.integers(0, 4))
    topk = int(rng.choice([1, 2, 5, 50]))
    ids = [matcher.upload_u8(img) for img in images]
    vid = None
    try:
        words = matcher.train_vocabulary(ids, nwords, iterations)
        ewords = orc.train_vocabulary(images, nwords, iterations) if sum(map(len, images)) else np.zeros((0, 128), np.uint8)
        np.testing.assert_array_equal(words, ewords)
        if len(words):
            vid = matcher.upload_u8(words)
            pairs, scores = matcher.shortlist_pairs(ids, vid, topk, want_scores=True)
            wl = [orc.knn2_u8(img, words)[0] if len(img) else np.zeros(0, np.int32) for img in images]
            epairs, escores = orc.shortlist(wl, len(words), topk)
            np.testing.assert_array_equal(scores, escores)
            np.testing.assert_array_equal(pairs, epairs)
    finally:
        for s in ids + ([vid] if vid is not None else []):
            matcher.release(s)
