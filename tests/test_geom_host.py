"""CPU checks around the geometric verification step: the oracle's RANSAC procedure recovers planted
two-view geometry, and the inlier records of a .mat file round-trip through the writers and readers
that stand in for MatchFile::WriteFMatch / GetIMatch / GetIMatchR.

The estimate itself has no reference output to be compared with (the reference calls OpenCV, which is
not installed): parity unpinned.  The record layout follows MatchFile.cpp:660-748 and :961-1084."""
import os
import struct

import numpy as np

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import synth


def _sampson(F, a, b):
    x0 = np.c_[a, np.ones(len(a))]
    x1 = np.c_[b, np.ones(len(b))]
    Fx0 = x0 @ F.T
    Ftx1 = x1 @ F
    e = (x1 * Fx0).sum(1)
    return np.abs(e) / np.sqrt(Fx0[:, 0] ** 2 + Fx0[:, 1] ** 2 + Ftx1[:, 0] ** 2 + Ftx1[:, 1] ** 2)


def test_oracle_recovers_planted_geometry():
    xy0, xy1, i0, i1, true, Ftrue = synth.two_view(600, seed=3, inlier_fraction=0.55, noise=0.4)
    F, in0, in1 = orc.geom_pair(xy0, xy1, i0, i1, max_sampson=2.0, hypotheses=512, seed=9)
    got = set(zip(in0.tolist(), in1.tolist()))
    want = set(zip(i0[true].tolist(), i1[true].tolist()))
    assert len(got & want) >= 0.97 * len(want)         # nearly all true correspondences are inliers
    assert len(got - want) <= 0.1 * len(want)          # few wrong pairings lie on an epipolar line by chance
    F = F.astype(np.float64)
    assert abs(np.linalg.norm(F) - 1.0) < 1e-5
    assert np.linalg.svd(F, compute_uv=False)[2] < 1e-6  # rank 2
    # the true correspondences satisfy the estimated geometry to about the noise level
    d = _sampson(F, xy0[i0[true]].astype(np.float64), xy1[i1[true]].astype(np.float64))
    assert np.median(d) < 1.0
    # lists come in match order
    pos = {(a, b): k for k, (a, b) in enumerate(zip(i0.tolist(), i1.tolist()))}
    ks = [pos[p] for p in zip(in0.tolist(), in1.tolist())]
    assert ks == sorted(ks)


def test_oracle_degenerate_inputs():
    xy0, xy1, i0, i1, _, _ = synth.two_view(40, seed=4)
    F, in0, in1 = orc.geom_pair(xy0, xy1, i0[:7], i1[:7])          # fewer than eight matches
    assert len(in0) == 0 and not F.any()
    same = np.zeros((40, 2), np.float32)                            # all keypoints coincide
    F, in0, in1 = orc.geom_pair(same, same, i0, i1)
    assert len(in0) == 0 and not F.any()
    bad = i0.copy()
    bad[::2] = 10**6                                                # out-of-range feature indices never count
    F, in0, in1 = orc.geom_pair(xy0, xy1, bad, i1, hypotheses=64)
    assert (in0 < 40).all()
    F, in0, in1 = orc.geom_pair(xy0, xy1, i0, i1, min_inliers=41)   # cannot be reached
    assert len(in0) == 0 and not F.any()


def test_inlier_records_round_trip(tmp_path):
    a, b = str(tmp_path / "imgA"), str(tmp_path / "sub" / "imgB")
    os.makedirs(os.path.dirname(b))
    rng = np.random.default_rng(5)
    nm = 37
    p0 = rng.permutation(500)[:nm].astype(np.int32)
    p1 = rng.permutation(400)[:nm].astype(np.int32)
    vsm.write_pmatch(a, b, 500, 400, p0, p1)
    sel = np.sort(rng.permutation(nm)[:21])
    F = rng.standard_normal((3, 3)).astype(np.float32)
    assert vsm.write_fmatch(a, b, p0[sel], p1[sel], F) == 1
    # putative lists are untouched
    g0, g1 = vsm.get_pmatch(a, b, 400)
    np.testing.assert_array_equal(g0, p0)
    np.testing.assert_array_equal(g1, p1)
    nm_a, Fa, r0, r1 = vsm.get_imatch(a, b, 400)
    assert nm_a == nm
    np.testing.assert_array_equal(Fa, F)
    np.testing.assert_array_equal(r0, p0[sel])
    np.testing.assert_array_equal(r1, p1[sel])
    # the other image's file holds the transposed matrix and the swapped rows ...
    nm_b, Fb, s0, s1 = vsm.get_imatch(b, a, 500)
    assert nm_b == nm
    np.testing.assert_array_equal(Fb, F.T)
    np.testing.assert_array_equal(s0, p1[sel])
    np.testing.assert_array_equal(s1, p0[sel])
    # ... which the reversed reader turns back
    nm_r, Fr, t0, t1 = vsm.get_imatch_r(b, a, 500)
    np.testing.assert_array_equal(Fr, F)
    np.testing.assert_array_equal(t0, p0[sel])
    np.testing.assert_array_equal(t1, p1[sel])
    assert vsm.get_match_count(a, b) == (nm, 21)
    # bytes: NM, record version 'MRV3', NF, then F row-major at offset 4 + 4 + 16 of the block
    raw = open(a + ".mat", "rb").read()
    loc = struct.unpack_from("<5i", raw, 20)
    blk = loc[1]
    assert struct.unpack_from("<i", raw, blk)[0] == nm
    assert raw[blk + 4:blk + 8] == b"MRV3"
    assert struct.unpack_from("<i", raw, blk + 8)[0] == 21
    np.testing.assert_array_equal(np.frombuffer(raw, np.float32, 9, blk + 4 + 4 + 16).reshape(3, 3), F)
    inl = np.frombuffer(raw, np.int32, 42, blk + 4 + 184 + nm * 8)
    np.testing.assert_array_equal(inl[:21], p0[sel])
    np.testing.assert_array_equal(inl[21:], p1[sel])
    # a wrong feature count is rejected as the reference does, and an unknown partner reports nothing
    assert vsm.get_imatch(a, b, 399)[0] == 0
    assert vsm.write_fmatch(a, str(tmp_path / "other"), p0[:1], p1[:1], F) == 0


def test_cached_writer_leaves_the_same_files(tmp_path):
    """vsm_mat_writer_put + put_fmatch over many pairs = one write_pmatch + write_fmatch call per pair."""
    rng = np.random.default_rng(6)
    n = 6
    fc = [int(x) for x in rng.integers(200, 400, n)]
    recs = []
    for a in range(n):
        for b in range(a + 1, n):
            nm = int(rng.integers(0, 60))
            p0 = np.sort(rng.permutation(fc[a])[:nm]).astype(np.int32)
            p1 = rng.permutation(fc[b])[:nm].astype(np.int32)
            sel = np.sort(rng.permutation(nm)[: nm // 2])
            recs.append((a, b, p0, p1, sel, rng.standard_normal((3, 3)).astype(np.float32)))
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir()
    d2.mkdir()
    for a, b, p0, p1, sel, F in recs:
        vsm.write_pmatch(str(d1 / str(a)), str(d1 / str(b)), fc[a], fc[b], p0, p1)
        vsm.write_fmatch(str(d1 / str(a)), str(d1 / str(b)), p0[sel], p1[sel], F)
    from visual_slam_amd import host
    with host.MatWriter() as w:
        for a, b, p0, p1, sel, F in recs:
            w.put(str(d2 / str(a)), str(d2 / str(b)), fc[a], fc[b], p0, p1)
            w.put_fmatch(str(d2 / str(a)), str(d2 / str(b)), p0[sel], p1[sel], F)
    for i in range(n):
        assert open(str(d1 / ("%d.mat" % i)), "rb").read() == open(str(d2 / ("%d.mat" % i)), "rb").read()
    a, b, p0, p1, sel, F = recs[3]
    nm, Fg, r0, r1 = vsm.get_imatch(str(d2 / str(a)), str(d2 / str(b)), fc[b])
    assert nm == len(p0) or len(p0) == 0


def _cams(K):
    return np.asarray([K[0, 0], K[1, 1], K[0, 2], K[1, 2]] * 2, np.float32)


def test_oracle_recovers_planted_pose():
    """F by RANSAC, then E = K1^T F K0 decomposed: the rotation within a fraction of a degree and the
    translation direction within a few degrees of the planted ones; every inlier in front of both cameras."""
    for seed in (3, 5, 8):
        xy0, xy1, i0, i1, true, _, K, R, t = synth.two_view(600, seed=seed, inlier_fraction=0.6, noise=0.4, with_pose=True)
        F, in0, in1 = orc.geom_pair(xy0, xy1, i0, i1, max_sampson=2.0, hypotheses=512, seed=9)
        Rg, Tg, ne = orc.pose_pair(xy0, xy1, in0, in1, F, _cams(K))
        assert ne >= 0.95 * len(in0)
        Rg = Rg.astype(np.float64)
        assert abs(np.linalg.det(Rg) - 1.0) < 1e-5 and np.allclose(Rg @ Rg.T, np.eye(3), atol=1e-5)
        ang = np.degrees(np.arccos(np.clip((np.trace(Rg.T @ R) - 1.0) / 2.0, -1.0, 1.0)))
        assert ang < 0.5
        assert abs(np.linalg.norm(Tg) - 1.0) < 1e-5
        tdir = np.degrees(np.arccos(np.clip(np.dot(Tg, t / np.linalg.norm(t)), -1.0, 1.0)))
        assert tdir < 5.0
    # no inliers, a zero matrix: no pose
    Rg, Tg, ne = orc.pose_pair(xy0, xy1, in0[:0], in1[:0], F, _cams(K))
    assert ne == 0 and not Rg.any() and not Tg.any()
    Rg, Tg, ne = orc.pose_pair(xy0, xy1, in0, in1, np.zeros(9, np.float32), _cams(K))
    assert ne == 0 and not Rg.any()


def test_pose_records_round_trip(tmp_path):
    """NE, R, T, F1, F2 of TwoViewGeometry through vsm_mat_write_ematch / vsm_mat_get_pose: the first image's
    file holds them as given, the second image's file the exchanged view (R^T, -R^T T, focal lengths
    swapped: SetGeometryR, MatchFile.h:121-141); the inlier records stay what write_fmatch writes."""
    rng = np.random.default_rng(21)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    p0 = np.sort(rng.choice(500, 60, replace=False)).astype(np.int32)
    p1 = rng.choice(400, 60, replace=False).astype(np.int32)
    vsm.write_pmatch(a, b, 500, 400, p0, p1)
    sel = np.arange(0, 60, 2)
    F = rng.standard_normal((3, 3)).astype(np.float32)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    R = (q * np.sign(np.linalg.det(q))).astype(np.float32)
    T = rng.standard_normal(3).astype(np.float32)
    T /= np.linalg.norm(T)
    assert vsm.write_ematch(a, b, p0[sel], p1[sel], F, 27, R, T, 1200.0, 900.0) == 1
    nm, ne, Ra, Ta, f12 = vsm.get_pose(a, b, 400)
    assert (nm, ne, f12) == (60, 27, (1200.0, 900.0))
    np.testing.assert_array_equal(Ra, R)
    np.testing.assert_array_equal(Ta, T)
    nm, ne, Rb, Tb, f12 = vsm.get_pose(b, a, 500)
    assert (nm, ne, f12) == (60, 27, (900.0, 1200.0))
    np.testing.assert_array_equal(Rb, R.T)
    np.testing.assert_allclose(Tb, -(R.T.astype(np.float64) @ T), rtol=0, atol=1e-6)
    # exchanging the view twice gives the pose back
    np.testing.assert_allclose(-(Rb.T.astype(np.float64) @ Tb), T, rtol=0, atol=1e-6)
    nm_a, Fa, r0, r1 = vsm.get_imatch(a, b, 400)
    np.testing.assert_array_equal(Fa, F)
    np.testing.assert_array_equal(r0, p0[sel])
    np.testing.assert_array_equal(r1, p1[sel])
    # record bytes: NF, NE at the head of the geometry, R after F
    raw = open(a + ".mat", "rb").read()
    loc = struct.unpack("<i", raw[20 + 4:20 + 8])[0]
    assert struct.unpack("<ii", raw[loc + 8:loc + 16]) == (30, 27)
    assert np.frombuffer(raw[loc + 8 + 16 + 36:loc + 8 + 16 + 72], np.float32).tolist() == R.reshape(9).tolist()
    # a pose without an F is refused, ne = 0 writes none
    assert vsm.lib().vsm_mat_write_ematch(a.encode(), b.encode(), 0, None, None, None, 5, None, None, 1.0, 1.0) < 0
    assert vsm.write_ematch(a, b, p0[sel], p1[sel], F, 0, R, T, 1.0, 2.0) == 1
    assert vsm.get_pose(a, b, 400)[1] == 0 and not vsm.get_pose(a, b, 400)[2].any()
