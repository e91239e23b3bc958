"""GPU parity: geometric verification of putative matches through the C ABI against the CPU oracle.

Bar: the fundamental matrix bit for bit (float32), inlier lists identical.  Every operation of the
procedure is an IEEE add / multiply / divide / square root in a fixed order, in double for the solve
and in float for the scoring, so the two implementations must agree exactly.  The estimate itself has
no output of the reference to be held against (it calls OpenCV): parity unpinned."""
import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu


def _scene(sizes, seed):
    sets, lists = [], []
    for k, n in enumerate(sizes):
        xy0, xy1, i0, i1, _, _ = synth.two_view(max(n, 1), seed=seed + k, inlier_fraction=0.7 + 0.04 * (k % 5))
        sets.append((xy0, xy1))
        lists.append((i0[:n], i1[:n]))
    return sets, lists


def _check(matcher, sets, lists, **kw):
    ids = [(matcher.upload_xy(a), matcher.upload_xy(b)) for a, b in sets]
    off = np.zeros(len(lists) + 1, np.int64)
    off[1:] = np.cumsum([len(l[0]) for l in lists])
    i0 = np.concatenate([l[0] for l in lists]) if lists else np.zeros(0, np.int32)
    i1 = np.concatenate([l[1] for l in lists]) if lists else np.zeros(0, np.int32)
    F, nf, in_off, g0, g1 = matcher.verify_matches(ids, off, i0, i1, **kw)
    total = 0
    for p, ((a, b), (l0, l1)) in enumerate(zip(sets, lists)):
        oF, o0, o1 = orc.geom_pair(a, b, l0, l1, pair_pos=p, **kw)
        np.testing.assert_array_equal(F[p], oF, err_msg="pair %d" % p)
        assert nf[p] == len(o0)
        np.testing.assert_array_equal(g0[in_off[p]:in_off[p + 1]], o0)
        np.testing.assert_array_equal(g1[in_off[p]:in_off[p + 1]], o1)
        total += len(o0)
    assert in_off[-1] == total
    for a, b in ids:
        matcher.release(a)
        matcher.release(b)
    return nf


def test_pairs_of_many_sizes(matcher):
    sizes = [0, 5, 8, 9, 100, 600, 3000, 64, 1, 257]
    sets, lists = _scene(sizes, seed=70)
    nf = _check(matcher, sets, lists, max_sampson=2.0, hypotheses=512, seed=11)
    assert nf[5] > 200 and nf[6] > 1000 and nf[0] == 0 and nf[1] == 0


def test_thresholds_seeds_and_sample_counts(matcher):
    sets, lists = _scene([400, 700], seed=80)
    for kw in (dict(max_sampson=0.5, hypotheses=100, seed=1), dict(max_sampson=4.0, hypotheses=1000, seed=2),
               dict(max_sampson=1.0, hypotheses=130, seed=3, min_inliers=10**6)):
        _check(matcher, sets, lists, **kw)


def test_degenerate_and_out_of_range(matcher):
    xy0, xy1, i0, i1, _, _ = synth.two_view(50, seed=90)
    same = np.zeros((50, 2), np.float32)
    bad = i0.copy()
    bad[::3] = 10**6
    neg = i1.copy()
    neg[5] = -1
    _check(matcher, [(same, same), (xy0, xy1), (xy0, xy1)], [(i0, i1), (bad, i1), (i0, neg)], hypotheses=64, seed=5)


def test_lists_of_the_last_submission_stay_on_the_device(matcher):
    """SIFT matching followed by verification of the device-resident match lists."""
    rng = np.random.default_rng(7)
    imgs, xys = [], []
    base_xy0, base_xy1, _, _, _, _ = synth.two_view(1500, seed=91, inlier_fraction=1.0, noise=0.3)
    a, b = synth.sift_pair(1500, 1500, seed=92, planted=0.5)
    c, _ = synth.sift_pair(900, 10, seed=93)
    sa, sb, sc = matcher.upload_u8(a), matcher.upload_u8(b), matcher.upload_u8(c)
    xc = rng.uniform(0, 1000, (900, 2)).astype(np.float32)
    pairs = [(sa, sb), (sa, sc), (sc, sb)]
    tot = matcher.match_pairs_device(pairs, ratio=0.8, mutual=True)
    off, i0, i1 = matcher.fetch_matches(len(pairs), tot)
    assert off[1] > 300
    xy = {sa: base_xy0, sb: base_xy1, sc: xc}
    ids = {s: matcher.upload_xy(v) for s, v in xy.items()}
    xp = [(ids[p], ids[q]) for p, q in pairs]
    F, nf, in_off, g0, g1 = matcher.verify_matches(xp, max_sampson=1.5, hypotheses=200, seed=4, cap=tot)
    for k, (p, q) in enumerate(pairs):
        oF, o0, o1 = orc.geom_pair(xy[p], xy[q], i0[off[k]:off[k + 1]], i1[off[k]:off[k + 1]], max_sampson=1.5,
                                   hypotheses=200, seed=4, pair_pos=k)
        np.testing.assert_array_equal(F[k], oF)
        np.testing.assert_array_equal(g0[in_off[k]:in_off[k + 1]], o0)
        np.testing.assert_array_equal(g1[in_off[k]:in_off[k + 1]], o1)
    # without the total the binding sizes its buffers from a first call
    F2, nf2, in_off2, h0, h1 = matcher.verify_matches(xp, max_sampson=1.5, hypotheses=200, seed=4)
    np.testing.assert_array_equal(F2, F)
    np.testing.assert_array_equal(h0, g0)
    for s in list(ids.values()) + [sa, sb, sc]:
        matcher.release(s)


def test_rejects_descriptor_sets_and_bad_arguments(matcher):
    a, b = synth.sift_pair(10, 10, seed=1)
    s = matcher.upload_u8(a)
    x = matcher.upload_xy(np.zeros((10, 2), np.float32))
    off = np.array([0, 0], np.int64)
    with pytest.raises(vsm.VsmError):
        matcher.verify_matches([(s, x)], off, np.zeros(0, np.int32), np.zeros(0, np.int32))
    with pytest.raises(vsm.VsmError):
        matcher.verify_matches([(x, x)], off, np.zeros(0, np.int32), np.zeros(0, np.int32), max_sampson=0.0)
    with pytest.raises(vsm.VsmError):
        matcher.match_pair(x, x)
    matcher.release(s)
    matcher.release(x)


def test_estimate_fundamental_helper(matcher):
    xy0, xy1, i0, i1, true, Ftrue = synth.two_view(500, seed=95, inlier_fraction=0.8, noise=0.4)
    got = vsm.estimate_fundamental(matcher, xy0[i0], xy1[i1], 1.5, samples=300, seed=2)
    assert got is not None
    F, mask = got
    assert mask.sum() >= 0.97 * true.sum() and (mask & ~true).sum() <= 0.05 * true.sum()
    oF, o0, _ = orc.geom_pair(xy0[i0], xy1[i1], np.arange(500), np.arange(500), max_sampson=1.5, hypotheses=300, seed=2)
    np.testing.assert_array_equal(F, oF)
    np.testing.assert_array_equal(np.nonzero(mask)[0], o0)
    assert vsm.estimate_fundamental(matcher, xy0[:7], xy1[:7], 1.5) is None


def test_pose_bit_exact_and_close_to_planted(matcher):
    """vsm_recover_pose against the oracle bit for bit (R, T in float32, the count), over pairs of many
    sizes, different cameras per image, a pair without inliers and a zero matrix; and within half a degree
    (rotation) / a few degrees (translation direction) of the planted pose where the scene has one."""
    sizes = [600, 0, 40, 1500, 9, 300]
    scenes = [synth.two_view(max(n, 1), seed=200 + k, inlier_fraction=0.7, noise=0.4, with_pose=True) for k, n in enumerate(sizes)]
    ids = [(matcher.upload_xy(s[0]), matcher.upload_xy(s[1])) for s in scenes]
    lists = [(s[2][:n], s[3][:n]) for s, n in zip(scenes, sizes)]
    off = np.zeros(len(sizes) + 1, np.int64)
    off[1:] = np.cumsum(sizes)
    F, nf, in_off, g0, g1 = matcher.verify_matches(ids, off, np.concatenate([l[0] for l in lists]),
                                                   np.concatenate([l[1] for l in lists]), max_sampson=2.0, hypotheses=512, seed=5)
    cams = np.zeros((len(sizes), 8), np.float32)
    for p, s in enumerate(scenes):
        K = s[6]
        cams[p] = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    cams[3, 4:] *= 1.0          # (same camera twice in the synthetic scenes)
    F2 = F.copy()
    F2[5] = 0.0                 # a zero matrix has no decomposition
    R, T, ne = matcher.recover_pose(ids, cams, F2, in_off, g0, g1)
    for p, s in enumerate(scenes):
        oR, oT, one = orc.pose_pair(s[0], s[1], g0[in_off[p]:in_off[p + 1]], g1[in_off[p]:in_off[p + 1]], F2[p], cams[p])
        np.testing.assert_array_equal(R[p], oR, err_msg="pair %d" % p)
        np.testing.assert_array_equal(T[p], oT, err_msg="pair %d" % p)
        assert ne[p] == one
    assert ne[1] == 0 and ne[5] == 0 and not R[5].any()
    for p in (0, 3):
        Rt, tt = scenes[p][7], scenes[p][8]
        ang = np.degrees(np.arccos(np.clip((np.trace(R[p].astype(np.float64).T @ Rt) - 1.0) / 2.0, -1.0, 1.0)))
        tdir = np.degrees(np.arccos(np.clip(np.dot(T[p], tt / np.linalg.norm(tt)), -1.0, 1.0)))
        assert ang < 0.5 and tdir < 5.0 and ne[p] >= 0.95 * nf[p]
    for a, b in ids:
        matcher.release(a)
        matcher.release(b)


def test_estimate_pose_in_the_reference_calling_style(matcher):
    """host.estimate_pose(kpts0, kpts1, K0, K1, thresh) -> (R, t, mask), None for fewer than five matches
    (models/utils.py:288-312)."""
    xy0, xy1, i0, i1, true, _, K, Rt, tt = synth.two_view(500, seed=31, inlier_fraction=0.7, noise=0.4, with_pose=True)
    k0, k1 = xy0[i0], xy1[i1]
    assert vsm.estimate_pose(matcher, k0[:4], k1[:4], K, K, 2.0) is None
    R, t, mask = vsm.estimate_pose(matcher, k0, k1, K, K, 2.0, samples=512, seed=3)
    assert mask.dtype == bool and len(mask) == 500 and (mask & true).sum() >= 0.95 * true.sum()
    ang = np.degrees(np.arccos(np.clip((np.trace(R.astype(np.float64).T @ Rt) - 1.0) / 2.0, -1.0, 1.0)))
    assert ang < 0.5
