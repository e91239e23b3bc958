"""GPU parity: float descriptors (SuperPoint-256 style) through the C ABI against the CPU oracle
and the reference-generated fixtures.

Bar: indices bit-exact; scores/distances bit-exact against the oracle (dot products: the k-ascending
fmaf chain, which the f32-input MFMA is; squared distances: the k-ascending fmaf chain of (a - b)^2 on the
two best rows), and within 1e-5 relative of a float64 computation, near-duplicates included."""
import math
import os

import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import _binding as B
from visual_slam_amd import host, synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(autouse=True, params=[vsm.F32_FAST_EXACT, vsm.F32_MFMA_F32], ids=["bf16+rerank", "f32mfma"])
def f32_mode(matcher, request):
    """Every test runs twice: bf16 coarse pass + exact re-rank, and the f32 matrix-core kernel alone."""
    matcher.set_f32_mode(request.param)
    yield request.param
    matcher.set_f32_mode(vsm.F32_FAST_EXACT)


@pytest.fixture(params=[vsm.F32_REVERSE_ALL, vsm.F32_REVERSE_CANDIDATES], ids=["both-directions", "candidates-only"])
def reverse_mode(matcher, request):
    """Pair tests run with the reverse searches of the mutual check over every train row, and over the
    candidates' train rows only (what large submissions with a selective rule do by themselves)."""
    matcher.set_f32_reverse(request.param)
    yield request.param
    matcher.set_f32_reverse(vsm.F32_REVERSE_AUTO)


def _knn_check(matcher, a, b, metric):
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    try:
        gi1, gi2, gk1, gk2 = matcher.knn2(s0, s1, metric, want_idx2=True)
        oi, ok1, ok2 = orc.knn2_f32(a, b, metric)
        np.testing.assert_array_equal(gi1, oi)
        np.testing.assert_array_equal(gk1, ok1)
        np.testing.assert_array_equal(gk2, ok2)
        if len(b) >= 2:
            assert (gi2 >= 0).all() and (gi2 != gi1).all()
    finally:
        matcher.release(s0)
        matcher.release(s1)


@pytest.mark.parametrize("n0,n1,dim", [(1, 1, 256), (3, 2, 256), (200, 300, 256), (257, 1025, 256),
                                       (1024, 1024, 256), (100, 90, 128), (70, 50, 64), (40, 33, 100), (9, 300, 7)])
@pytest.mark.parametrize("metric", [vsm.METRIC_DOT, vsm.METRIC_L2])
def test_knn2_matches_oracle(matcher, n0, n1, dim, metric):
    a, b = synth.superpoint_pair(n0, n1, dim, seed=100 + n0 + n1 + dim)
    _knn_check(matcher, a, b, metric)


@pytest.mark.parametrize("n0,n1,dim", [(4100, 5000, 256), (4096, 4096, 128), (4200, 9000, 64)])
@pytest.mark.parametrize("metric", [vsm.METRIC_DOT, vsm.METRIC_L2])
def test_one_search_cut_into_many_splits(matcher, n0, n1, dim, metric):
    """A few thousand queries against a few thousand rows: the search is cut into 16 to 32 splits to fill the chip and
    the re-rank reads, per query, only the lists of the splits that can hold one of its two best rows."""
    a, b = synth.superpoint_pair(n0, n1, dim, seed=7 + dim)
    _knn_check(matcher, a, b, metric)


@pytest.mark.parametrize("metric", [vsm.METRIC_DOT, vsm.METRIC_L2])
def test_many_splits_with_thousands_of_equal_rows(matcher, metric):
    """3000 copies of one row among 5000: for the queries equal to it every copy survives the coarse pass (lists longer
    than one trip, more rows than a group's queue holds); the answer is still the two lowest indices."""
    rng = np.random.default_rng(11)
    a, b = synth.superpoint_pair(4100, 5000, 256, seed=12)
    v = b[17].copy()
    copies = rng.choice(5000, 3000, replace=False)
    b[copies] = v
    a[rng.choice(4100, 12, replace=False)] = v
    _knn_check(matcher, a, b, metric)


def test_knn2_non_unit_vectors_l2(matcher):
    rng = np.random.default_rng(5)
    a = rng.random((300, 256), dtype=np.float32) * 3
    b = rng.random((500, 256), dtype=np.float32) * 3
    _knn_check(matcher, a, b, vsm.METRIC_L2)
    D = ((a[:, None, :].astype(np.float64) - b[None].astype(np.float64)) ** 2).sum(-1)
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    gi, gd1, gd2 = matcher.knn2(s0, s1, vsm.METRIC_L2)
    np.testing.assert_array_equal(gi, D.argmin(1))
    np.testing.assert_allclose(gd1, D.min(1), rtol=1e-5, atol=0)
    np.testing.assert_allclose(gd2, np.sort(D, 1)[:, 1], rtol=1e-5, atol=0)
    matcher.release(s0)
    matcher.release(s1)


def test_ties_first_index(matcher):
    rng = np.random.default_rng(6)
    base = synth.unit_f32(30, 256, rng)
    a = np.concatenate([base, base[:10]])
    b = np.concatenate([base[::-1], base[:20], base])
    for metric in (vsm.METRIC_DOT, vsm.METRIC_L2):
        _knn_check(matcher, a, b, metric)


def test_two_best_rows_inside_one_group_of_a_tile(matcher):
    """A lane of the coarse pass keeps its runner-up over tile maxima (the 16 rows of a tile it holds), not over
    rows: when a query's two best rows are among those 16 rows its threshold stays below the runner-up over rows,
    and both rows must still come out, in order.  Rows 1 and 2 of a tile belong to one lane, rows 1 and 5 to two."""
    rng = np.random.default_rng(77)
    nq = 300
    a = synth.unit_f32(nq, 256, rng)
    b = synth.unit_f32(nq * 32 + 40, 256, rng)
    want1 = np.empty(nq, np.int32)
    want2 = np.empty(nq, np.int32)
    for q in range(nq):
        j1, j2 = 32 * q + 1, 32 * q + (2 if q % 2 == 0 else 5)
        if q % 3 == 0:
            j1, j2 = j2, j1          # the better row may come second in the tile
        for j, noise in ((j1, 0.01), (j2, 0.03)):
            v = a[q] + noise * rng.standard_normal(256).astype(np.float32)
            b[j] = v / np.linalg.norm(v)
        want1[q], want2[q] = j1, j2
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    try:
        for metric in (vsm.METRIC_DOT, vsm.METRIC_L2):
            gi1, gi2, gk1, gk2 = matcher.knn2(s0, s1, metric, want_idx2=True)
            oi, ok1, ok2 = orc.knn2_f32(a, b, metric)
            np.testing.assert_array_equal(oi, want1)
            np.testing.assert_array_equal(gi1, want1)
            np.testing.assert_array_equal(gi2, want2)
            np.testing.assert_array_equal(gk1, ok1)
            np.testing.assert_array_equal(gk2, ok2)
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_layout_dim_major(matcher):
    a, b = synth.superpoint_pair(150, 170, 256, seed=7)
    s0 = matcher.upload_f32(np.ascontiguousarray(a.T), vsm.LAYOUT_DIM_MAJOR)
    s1 = matcher.upload_f32(np.ascontiguousarray(b.T), vsm.LAYOUT_DIM_MAJOR)
    gi, gk1, gk2 = matcher.knn2(s0, s1, vsm.METRIC_DOT)
    oi, ok1, ok2 = orc.knn2_f32(a, b, orc.METRIC_DOT)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gk1, ok1)
    matcher.release(s0)
    matcher.release(s1)


@pytest.mark.parametrize("metric,ratio", [(vsm.METRIC_DOT, None), (vsm.METRIC_L2, 0.8), (vsm.METRIC_L2, None)])
def test_match_dense_matches_oracle(matcher, metric, ratio, reverse_mode):
    a, b = synth.superpoint_pair(600, 700, 256, seed=8)
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    fr = (4, 5) if ratio else None
    for mutual in (True, False):
        g0, g1 = matcher.match_pair_dense(s0, s1, ratio=ratio, mutual=mutual, metric=metric, min_score=0.5)
        o0, o1, _, _ = orc.match_dense_f32(a, b, metric, orc.params(fr, mutual, min_score=0.5))
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
        i0, i1, sc = matcher.match_pair(s0, s1, ratio=ratio, mutual=mutual, metric=metric, min_score=0.5,
                                        return_dist=True)
        np.testing.assert_array_equal(i0, np.nonzero(o0 >= 0)[0])
        np.testing.assert_array_equal(i1, o0[o0 >= 0])
    assert (o0 >= 0).sum() > 50
    matcher.release(s0)
    matcher.release(s1)


def test_mutual_block_fixture(matcher, reverse_mode):
    """Same outputs as the reference's mutual-argmax block run on the same descriptors."""
    z = np.load(os.path.join(G, "mutual_block.npz"))
    for tag in ("a", "b"):
        out = host.mutual_nn_match(matcher, z["desc0_" + tag], z["desc1_" + tag], match_threshold=0.2)
        np.testing.assert_array_equal(out["matches0"], z["matches0_" + tag])
        np.testing.assert_array_equal(out["matches1"], z["matches1_" + tag])
        np.testing.assert_allclose(out["matching_scores0"], z["mscores0_" + tag], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out["matching_scores1"], z["mscores1_" + tag], rtol=1e-5, atol=1e-6)
        # the library's own dense matcher with the threshold moved onto the raw dot product
        s0 = matcher.upload_f32(z["desc0_" + tag], vsm.LAYOUT_DIM_MAJOR)
        s1 = matcher.upload_f32(z["desc1_" + tag], vsm.LAYOUT_DIM_MAJOR)
        m0, m1 = matcher.match_pair_dense(s0, s1, ratio=None, mutual=True, metric=vsm.METRIC_DOT,
                                          min_score=np.float32(16.0 * math.log(0.2)))
        np.testing.assert_array_equal(m0, z["matches0_" + tag])
        np.testing.assert_array_equal(m1, z["matches1_" + tag])
        matcher.release(s0)
        matcher.release(s1)


def test_nearest_embed_fixture(matcher):
    z = np.load(os.path.join(G, "nearest_embed.npz"))
    for sfx in ("", "2"):
        res, am = host.nearest_embed(matcher, z["x" + sfx], z["emb" + sfx])
        np.testing.assert_array_equal(am, z["argmin" + sfx])
        np.testing.assert_array_equal(res, z["result" + sfx])
        ores, oam = orc.nearest_embed_f32(z["x" + sfx], z["emb" + sfx])
        np.testing.assert_array_equal(am, oam)


def test_nearest_embed_near_ties_fixture(matcher):
    """Codes in pairs 1e-6 apart, points 1e-4 .. 1e-2 from one of them: the argmin must be the one the
    reference's nearest_embed returned (tests/golden/gen_golden.py), which ranks by the norm of the difference."""
    z = np.load(os.path.join(G, "nearest_embed_near_ties.npz"))
    res, am = host.nearest_embed(matcher, z["x"], z["emb"])
    np.testing.assert_array_equal(am, z["argmin"])
    np.testing.assert_array_equal(res, z["result"])


def test_near_duplicates_l2_relative_tolerance(matcher):
    """Near-duplicates at 1e-5 .. 1e-1: distances within 1e-5 RELATIVE of float64, no absolute floor; and the
    ratio test and mutual check on them equal the oracle's."""
    a, b = synth.superpoint_pair(400, 600, 256, seed=12)
    rng = np.random.default_rng(13)
    for r, eps in zip(range(0, 200, 4), np.logspace(-5, -1, 50)):
        b[r + 1] = a[r] + (eps / 16.0 * rng.standard_normal(256)).astype(np.float32)
        b[r + 2] = b[r + 1] + (1e-6 / 16.0 * rng.standard_normal(256)).astype(np.float32)   # two rows 1e-6 apart
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    try:
        gi, gi2, gd1, gd2 = matcher.knn2(s0, s1, vsm.METRIC_L2, want_idx2=True)
        D = ((a[:, None, :].astype(np.float64) - b[None].astype(np.float64)) ** 2).sum(-1)
        np.testing.assert_allclose(gd1, D.min(1), rtol=1e-5, atol=0)
        np.testing.assert_allclose(gd2, np.sort(D, 1)[:, 1], rtol=1e-5, atol=0)
        oi, od1, od2 = orc.knn2_f32(a, b, orc.METRIC_L2)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd1, od1)
        np.testing.assert_array_equal(gd2, od2)
        for mutual in (True, False):
            g0, g1 = matcher.match_pair_dense(s0, s1, ratio=0.8, mutual=mutual, metric=vsm.METRIC_L2)
            o0, o1, _, _ = orc.match_dense_f32(a, b, orc.METRIC_L2, orc.params((4, 5), mutual))
            np.testing.assert_array_equal(g0, o0)
            np.testing.assert_array_equal(g1, o1)
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_database_slices_merge(matcher):
    """Database cut into contiguous slices, per-slice 2-NN merged = 2-NN over the whole database."""
    rng = np.random.default_rng(9)
    q = synth.unit_f32(300, 256, rng)
    db = synth.unit_f32(2000, 256, rng)
    db[1500] = db[20]          # duplicate rows in different slices
    db[700] = q[5]
    db[1900] = q[5]
    cuts = [0, 512, 1100, 1100 + 333, 2000]
    sq = matcher.upload_f32(q)
    for metric in (vsm.METRIC_L2, vsm.METRIC_DOT):
        k1s, k2s, i1s, i2s = [], [], [], []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sd = matcher.upload_f32(db[lo:hi])
            i1, i2, k1, k2 = matcher.knn2(sq, sd, metric, want_idx2=True)
            k1s.append(k1.copy()); k2s.append(k2.copy()); i1s.append(i1.copy()); i2s.append(i2.copy())
            matcher.release(sd)
        ok1, ok2, oi1, oi2 = B.merge_top2_f32(np.stack(k1s), np.stack(k2s), np.stack(i1s), np.stack(i2s),
                                              np.asarray(cuts, np.int64), metric)
        sd = matcher.upload_f32(db)
        wi1, wi2, wk1, wk2 = matcher.knn2(sq, sd, metric, want_idx2=True)
        matcher.release(sd)
        np.testing.assert_array_equal(oi1, wi1)
        np.testing.assert_array_equal(oi2, wi2)
        np.testing.assert_array_equal(ok1, wk1)
        np.testing.assert_array_equal(ok2, wk2)
        oi, o1, o2 = orc.knn2_f32(q, db, metric)
        np.testing.assert_array_equal(oi1, oi)
        np.testing.assert_array_equal(ok1, o1)
        np.testing.assert_array_equal(ok2, o2)
    matcher.release(sq)


def test_fast_path_is_the_one_that_runs(matcher, f32_mode):
    a, b = synth.superpoint_pair(3000, 20000, 256, seed=31)
    for metric in (vsm.METRIC_DOT, vsm.METRIC_L2):
        _knn_check(matcher, a, b, metric)
        assert matcher.last_f32_path() == f32_mode


def test_scaled_and_clustered_vectors(matcher):
    """Norms far from 1 and tight clusters: the coarse margin scales with |a| max|b|."""
    rng = np.random.default_rng(32)
    centers = synth.unit_f32(40, 256, rng)
    b = (centers[rng.integers(0, 40, 6000)] + 0.02 * rng.standard_normal((6000, 256))).astype(np.float32)
    a = (centers[rng.integers(0, 40, 900)] + 0.02 * rng.standard_normal((900, 256))).astype(np.float32)
    b *= rng.uniform(0.01, 30.0, (6000, 1)).astype(np.float32)
    a *= rng.uniform(0.5, 900.0, (900, 1)).astype(np.float32)
    a[17] = 0.0
    for metric in (vsm.METRIC_DOT, vsm.METRIC_L2):
        _knn_check(matcher, a, b, metric)


def test_survivor_overflow_falls_back(matcher, f32_mode):
    """600000 copies of one row tie on every query, and every database tile leaves a list entry: the
    survivor lists overflow and the exact kernel takes over.  A sample of the queries is checked against
    the oracle."""
    rng = np.random.default_rng(33)
    a = synth.unit_f32(2000, 256, rng)
    b = np.repeat(synth.unit_f32(1, 256, rng), 600000, axis=0)
    b[4000] = a[3]
    sample = np.r_[0:24, 1990:2000]
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    try:
        for metric in (vsm.METRIC_DOT, vsm.METRIC_L2):
            gi1, gi2, gk1, gk2 = matcher.knn2(s0, s1, metric, want_idx2=True)
            assert matcher.last_f32_path() == vsm.F32_MFMA_F32
            oi, ok1, ok2 = orc.knn2_f32(a[sample], b, metric)
            np.testing.assert_array_equal(gi1[sample], oi)
            np.testing.assert_array_equal(gk1[sample], ok1)
            np.testing.assert_array_equal(gk2[sample], ok2)
            assert gi1[3] == 4000 and gi2[3] == 0
            assert np.isin(gi1, (0, 4000)).all() and np.isin(gi2, (0, 1)).all()   # ties go to the first row
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_full_survivor_list_drops_stale_entries(matcher, f32_mode):
    """Unit rows whose score for query 0 rises steadily inside every 1024-row stretch of the database:
    each row is appended when it is seen (four times what a list holds), but all except the last few
    fall below the risen threshold, so the full list is pruned in place and the search stays on the
    fast path."""
    rng = np.random.default_rng(34)
    a = synth.unit_f32(4096, 256, rng)
    r = rng.standard_normal((65536, 256))
    r -= (r @ a[0].astype(np.float64))[:, None] * a[0]          # random directions orthogonal to query 0
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    c = (-0.9 + 1.8 * (np.arange(65536) % 1024) / 1024.0)[:, None]
    b = (c * a[0] + np.sqrt(1.0 - c * c) * r).astype(np.float32)
    _knn_check(matcher, a, b, vsm.METRIC_DOT)
    assert matcher.last_f32_path() == f32_mode


@pytest.mark.parametrize("metric,ratio,mutual,dim", [(vsm.METRIC_DOT, None, True, 256), (vsm.METRIC_L2, 0.8, True, 256),
                                                     (vsm.METRIC_L2, 0.9, False, 256), (vsm.METRIC_DOT, None, True, 64),
                                                     (vsm.METRIC_L2, 0.8, True, 100)])
def test_float_pairs_in_one_submission(matcher, metric, ratio, mutual, dim, reverse_mode):
    """Many float pairs per call (ragged sizes, an empty set, a one-row set) against the oracle pair by
    pair.  The submission is large enough for the eight-queries-per-wave re-rank kernel; single pairs
    elsewhere in this file run the wave-per-query one."""
    rng = np.random.default_rng(41)
    sizes = [700, 513, 0, 900, 1, 1100, 256]
    pool = synth.unit_f32(600, dim, rng)
    imgs = []
    for n in sizes:
        d = synth.unit_f32(max(n, 1), dim, rng)[:n]
        k = min(n, 600) // 2
        if k:
            noisy = pool[rng.choice(600, k, replace=False)] + 0.8 / math.sqrt(dim) * rng.standard_normal((k, dim)).astype(np.float32)
            d[rng.choice(n, k, replace=False)] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
        imgs.append(np.ascontiguousarray(d, np.float32))
    ids = [matcher.upload_f32(d) for d in imgs]
    pairs = [(a, b) for a in range(len(sizes)) for b in range(len(sizes)) if a != b][:30]
    pr = np.asarray([[ids[a], ids[b]] for a, b in pairs], np.int32)
    off, i0, i1 = matcher.match_pairs(pr, ratio=ratio, mutual=mutual, metric=metric, min_score=0.5)
    fr = None if ratio is None else tuple(B.ratio_to_fraction(ratio))
    total = 0
    for k, (a, b) in enumerate(pairs):
        o0, _, _, _ = orc.match_dense_f32(imgs[a], imgs[b], metric, orc.params(fr, mutual, min_score=0.5))
        want0 = np.nonzero(o0 >= 0)[0]
        np.testing.assert_array_equal(i0[off[k]:off[k + 1]], want0)
        np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o0[want0])
        total += len(want0)
    assert off[-1] == total and total > 500
    for sid in ids:
        matcher.release(sid)


def test_database_query_full_size_both_paths_agree(matcher, f32_mode):
    """16384 queries against 1 M rows (the localisation-query shape): the bf16 + re-rank path and the
    f32 matrix-core kernel must return the same bits; a 64-query sample is checked against the oracle."""
    if f32_mode != vsm.F32_FAST_EXACT:
        pytest.skip("one comparison covers both paths")
    rng = np.random.default_rng(51)
    q = synth.unit_f32(16384, 256, rng)
    db = rng.standard_normal((1000000, 256), dtype=np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    db[777777] = q[5]
    db[12] = q[5]
    sq, sd = matcher.upload_f32(q), matcher.upload_f32(db)
    try:
        fast = matcher.knn2(sq, sd, vsm.METRIC_L2, want_idx2=True)
        assert matcher.last_f32_path() == vsm.F32_FAST_EXACT
        matcher.set_f32_mode(vsm.F32_MFMA_F32)
        exact = matcher.knn2(sq, sd, vsm.METRIC_L2, want_idx2=True)
        assert matcher.last_f32_path() == vsm.F32_MFMA_F32
        for f, e in zip(fast, exact):
            np.testing.assert_array_equal(f, e)
        assert fast[0][5] == 12 and fast[1][5] == 777777
        oi, ok1, ok2 = orc.knn2_f32(q[:64], db, orc.METRIC_L2)
        np.testing.assert_array_equal(fast[0][:64], oi)
        np.testing.assert_array_equal(fast[2][:64], ok1)
        np.testing.assert_array_equal(fast[3][:64], ok2)
    finally:
        matcher.set_f32_mode(vsm.F32_FAST_EXACT)
        matcher.release(sq)
        matcher.release(sd)


def test_device_merge_equals_host_merge(matcher):
    """Per-slice results left on the device and merged by the kernel = the host merge = the whole database."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")          # the runtime the library already loaded
    rng = np.random.default_rng(61)
    q = synth.unit_f32(500, 256, rng)
    db = synth.unit_f32(3000, 256, rng)
    db[2500] = db[20]
    db[700] = q[5]
    db[2900] = q[5]
    cuts = [0, 512, 1100, 1100, 2000, 3000]      # one empty slice
    ns, nq = len(cuts) - 1, len(q)
    sq = matcher.upload_f32(q)
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), C.c_size_t(ns * 4 * nq * 4)) == 0
    try:
        for metric in (vsm.METRIC_L2, vsm.METRIC_DOT):
            for s_, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
                sd = matcher.upload_f32(db[lo:hi])
                matcher.knn2_device(sq, sd, metric, dev.value + s_ * 4 * nq * 4)
                matcher.release(sd)
            k1, k2, i1, i2 = matcher.merge_top2_device(dev.value, np.asarray(cuts, np.int64), ns, nq, metric)
            h = np.empty((ns, 4, nq), np.int32)
            assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), dev, C.c_size_t(h.nbytes), 2) == 0
            hk1, hk2, hi1, hi2 = B.merge_top2_f32(h[:, 0].view(np.float32), h[:, 1].view(np.float32), h[:, 2], h[:, 3],
                                                  np.asarray(cuts, np.int64), metric)
            np.testing.assert_array_equal(i1, hi1)
            np.testing.assert_array_equal(i2, hi2)
            np.testing.assert_array_equal(k1, hk1)
            np.testing.assert_array_equal(k2, hk2)
            oi, o1, o2 = orc.knn2_f32(q, db, metric)
            np.testing.assert_array_equal(i1, oi)
            np.testing.assert_array_equal(k1, o1)
            np.testing.assert_array_equal(k2, o2)
    finally:
        hip.hipFree(dev)
        matcher.release(sq)


def _float_images(rng, sizes, dim):
    pool = synth.unit_f32(300, dim, rng)
    imgs = []
    for n in sizes:
        d = synth.unit_f32(max(n, 1), dim, rng)[:n]
        k = min(n, 300) // 2
        if k:
            noisy = pool[rng.choice(300, k, replace=False)] + 0.8 / math.sqrt(dim) * rng.standard_normal((k, dim)).astype(np.float32)
            d[rng.choice(n, k, replace=False)] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
        imgs.append(np.ascontiguousarray(d, np.float32))
    return imgs


def _check_float_pair_lists(imgs, pairs, off, i0, i1, metric, fr, mutual):
    total = 0
    for k, (a, b) in enumerate(pairs):
        o0, _, _, _ = orc.match_dense_f32(imgs[a], imgs[b], metric, orc.params(fr, mutual, min_score=0.5))
        want0 = np.nonzero(o0 >= 0)[0]
        np.testing.assert_array_equal(i0[off[k]:off[k + 1]], want0)
        np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o0[want0])
        total += len(want0)
    assert off[-1] == total
    return total


def test_float_pair_list_cut_by_scratch_budget(matcher, reverse_mode):
    """A float pair list that does not fit one submission's scratch: match_pairs cuts it and runs the
    submissions one after the other; the lists equal the oracle's pair by pair."""
    rng = np.random.default_rng(71)
    sizes = [700, 513, 900, 300, 1100]
    imgs = _float_images(rng, sizes, 256)
    ids = [matcher.upload_f32(d) for d in imgs]
    pairs = [(a, b) for a in range(len(sizes)) for b in range(len(sizes)) if a != b]
    pr = np.asarray([[ids[a], ids[b]] for a, b in pairs], np.int32)
    matcher.set_scratch_budget(8 << 20)       # a few pairs per submission at most
    try:
        for metric, ratio, mutual in ((vsm.METRIC_DOT, None, True), (vsm.METRIC_L2, 0.8, True)):
            off, i0, i1 = matcher.match_pairs(pr, ratio=ratio, mutual=mutual, metric=metric, min_score=0.5)
            fr = None if ratio is None else tuple(B.ratio_to_fraction(ratio))
            assert _check_float_pair_lists(imgs, pairs, off, i0, i1, metric, fr, mutual) > 200
    finally:
        matcher.set_scratch_budget(8 << 30)
        for sid in ids:
            matcher.release(sid)


def test_more_float_pairs_than_one_submission_takes(matcher):
    """More than 8192 tiny float pairs in one call: cut by the pair-count limit of a submission."""
    rng = np.random.default_rng(72)
    sizes = [40, 33, 57, 1, 64]
    imgs = _float_images(rng, sizes, 64)
    ids = [matcher.upload_f32(d) for d in imgs]
    base = [(a, b) for a in range(len(sizes)) for b in range(len(sizes)) if a != b]
    pairs = (base * (9000 // len(base) + 1))[:9000]
    pr = np.asarray([[ids[a], ids[b]] for a, b in pairs], np.int32)
    off, i0, i1 = matcher.match_pairs(pr, ratio=None, mutual=True, metric=vsm.METRIC_DOT, min_score=0.5)
    want = {}
    for k, (a, b) in enumerate(pairs):
        if (a, b) not in want:
            o0, _, _, _ = orc.match_dense_f32(imgs[a], imgs[b], vsm.METRIC_DOT, orc.params(None, True, min_score=0.5))
            sel = np.nonzero(o0 >= 0)[0]
            want[(a, b)] = (sel, o0[sel])
        np.testing.assert_array_equal(i0[off[k]:off[k + 1]], want[(a, b)][0])
        np.testing.assert_array_equal(i1[off[k]:off[k + 1]], want[(a, b)][1])
    for sid in ids:
        matcher.release(sid)


def test_config2_shape_4096_mutual(matcher, reverse_mode):
    """2 images x 4096 SuperPoint-256 descriptors, mutual nearest neighbour on the dot product: the
    launch the single-pair benchmark times (two searches + accept), dense and list forms."""
    a, b = synth.superpoint_pair(4096, 4096, 256, seed=77)
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    try:
        o0, o1, _, _ = orc.match_dense_f32(a, b, orc.METRIC_DOT, orc.params(None, True))
        g0, g1 = matcher.match_pair_dense(s0, s1, ratio=None, mutual=True, metric=vsm.METRIC_DOT)
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
        i0, i1 = matcher.match_pair(s0, s1, ratio=None, mutual=True, metric=vsm.METRIC_DOT)
        np.testing.assert_array_equal(i0, np.nonzero(o0 >= 0)[0])
        np.testing.assert_array_equal(i1, o0[o0 >= 0])
        assert (o0 >= 0).sum() > 1000
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_large_submission_searches_the_candidates_only(matcher, f32_mode):
    """A submission of the size the pair-list benchmark times (20 pairs of 4096 x 4096, dot product, score
    threshold, mutual): left to itself the library runs the reverse searches over the candidates' train rows
    only; the lists equal those of the both-directions form, and the oracle's on two of the pairs."""
    rng = np.random.default_rng(91)
    pool = synth.unit_f32(4096, 256, rng)
    imgs = []
    for _ in range(7):
        d = synth.unit_f32(4096, 256, rng)
        k = 1024
        noisy = pool[rng.choice(4096, k, replace=False)] + 0.05 * rng.standard_normal((k, 256)).astype(np.float32)
        d[rng.choice(4096, k, replace=False)] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
        imgs.append(np.ascontiguousarray(d, np.float32))
    ids = [matcher.upload_f32(d) for d in imgs]
    pairs = [(a, b) for a in range(7) for b in range(a + 1, 7)][:20]
    pr = np.asarray([[ids[a], ids[b]] for a, b in pairs], np.int32)
    try:
        off, i0, i1 = matcher.match_pairs(pr, ratio=None, mutual=True, metric=vsm.METRIC_DOT, min_score=0.5)
        if f32_mode == vsm.F32_FAST_EXACT:
            assert matcher.last_f32_reverse() == vsm.F32_REVERSE_CANDIDATES
        matcher.set_f32_reverse(vsm.F32_REVERSE_ALL)
        off2, j0, j1 = matcher.match_pairs(pr, ratio=None, mutual=True, metric=vsm.METRIC_DOT, min_score=0.5)
        if f32_mode == vsm.F32_FAST_EXACT:
            assert matcher.last_f32_reverse() == vsm.F32_REVERSE_ALL
        np.testing.assert_array_equal(off, off2)
        np.testing.assert_array_equal(i0, j0)
        np.testing.assert_array_equal(i1, j1)
        for k in (0, 19):
            a, b = pairs[k]
            o0, _, _, _ = orc.match_dense_f32(imgs[a], imgs[b], orc.METRIC_DOT, orc.params(None, True, min_score=0.5))
            sel = np.nonzero(o0 >= 0)[0]
            np.testing.assert_array_equal(i0[off[k]:off[k + 1]], sel)
            np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o0[sel])
        assert off[-1] > 20 * 100
    finally:
        matcher.set_f32_reverse(vsm.F32_REVERSE_AUTO)
        for sid in ids:
            matcher.release(sid)


@pytest.mark.parametrize("dim", [64, 128, 256])
@pytest.mark.parametrize("ncand", [1, 31, 33, 127, 129, 700])
def test_contested_columns_with_the_reverse_search_over_the_candidates_rows(matcher, f32_mode, dim, ncand):
    """Dot product + score threshold + mutual, candidates-only reverse searches: the reverse search of a pair
    looks at the candidates' rows of image 0 only.  `ncand` rows of image 0 have a partner above the threshold
    (list lengths either side of the 32-row tiles and of the 16 KiB steps at every dimension), and many of them
    contest the same train row: groups of query rows are noisy copies of ONE train row, so every column of a group
    has several candidates of which only the best (lowest row on exact ties: some copies are exact duplicates)
    survives.  Lists equal the both-directions form and the oracle's."""
    rng = np.random.default_rng(1000 * dim + ncand)
    n0, n1 = 900, 777
    a = synth.unit_f32(n0, dim, rng)
    b = synth.unit_f32(n1, dim, rng)
    rows = rng.choice(n0, ncand, replace=False)
    cols = rng.choice(n1, max(1, ncand // 3), replace=False)
    for k, r in enumerate(rows):
        v = b[cols[k % len(cols)]] + (0.0 if k % 5 == 0 else 0.04) * rng.standard_normal(dim).astype(np.float32)
        a[r] = v / np.linalg.norm(v)
    a = np.ascontiguousarray(a, np.float32)
    s0, s1 = matcher.upload_f32(a), matcher.upload_f32(b)
    pr = np.asarray([[s0, s1], [s1, s0], [s0, s0]], np.int32)
    try:
        lists = {}
        for mode in (vsm.F32_REVERSE_CANDIDATES, vsm.F32_REVERSE_ALL):
            matcher.set_f32_reverse(mode)
            lists[mode] = matcher.match_pairs(pr, ratio=None, mutual=True, metric=vsm.METRIC_DOT, min_score=0.75)
        for x, y in zip(lists[vsm.F32_REVERSE_CANDIDATES], lists[vsm.F32_REVERSE_ALL]):
            np.testing.assert_array_equal(x, y)
        off, i0, i1 = lists[vsm.F32_REVERSE_CANDIDATES]
        for k, (x, y) in enumerate([(a, b), (b, a), (a, a)]):
            o0, _, _, _ = orc.match_dense_f32(x, y, orc.METRIC_DOT, orc.params(None, True, min_score=0.75))
            sel = np.nonzero(o0 >= 0)[0]
            np.testing.assert_array_equal(i0[off[k]:off[k + 1]], sel)
            np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o0[sel])
        assert 0 < off[1] <= len(cols)          # one survivor per contested column at most
        assert off[3] - off[2] >= n0 - ncand    # a set against itself: every row whose only duplicate is itself
    finally:
        matcher.set_f32_reverse(vsm.F32_REVERSE_AUTO)
        matcher.release(s0)
        matcher.release(s1)
