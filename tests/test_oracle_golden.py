"""Pins the CPU oracle: against fixtures produced by running the reference's own Python code
(tests/golden/gen_golden.py), against byte sizes observed from the reference's C++ I/O classes,
and against plain numpy brute force."""
import math
import os

import numpy as np

import oracle_lib as orc
from visual_slam_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_nearest_embed_matches_reference():
    z = np.load(os.path.join(G, "nearest_embed.npz"))
    for sfx in ("", "2"):
        res, am = orc.nearest_embed_f32(z["x" + sfx], z["emb" + sfx])
        np.testing.assert_array_equal(am, z["argmin" + sfx])
        np.testing.assert_array_equal(res, z["result" + sfx])


def test_mutual_block_matches_reference():
    z = np.load(os.path.join(G, "mutual_block.npz"))
    for tag in ("a", "b"):
        d0, d1 = z["desc0_" + tag].T.copy(), z["desc1_" + tag].T.copy()  # [n][256]
        # exp(score / 16) > 0.2  <=>  dot > 16 * ln(0.2)
        p = orc.params(None, True, min_score=np.float32(16.0 * math.log(0.2)))
        m0, m1, b0, b1 = orc.match_dense_f32(d0, d1, orc.METRIC_DOT, p)
        np.testing.assert_array_equal(m0, z["matches0_" + tag])
        np.testing.assert_array_equal(m1, z["matches1_" + tag])
        i0, s0, _ = orc.knn2_f32(d0, d1, orc.METRIC_DOT)
        np.testing.assert_array_equal(i0, z["argmax0_" + tag])
        ci, _ = orc.colbest_f32(d0, d1, orc.METRIC_DOT)
        np.testing.assert_array_equal(ci, z["argmax1_" + tag])
        ms0 = np.where(m0 >= 0, np.exp(s0 / np.float32(16.0)), 0).astype(np.float32)
        # torch's blocked GEMM and the k-ordered fma chain differ in the last bits only
        np.testing.assert_allclose(ms0, z["mscores0_" + tag], rtol=1e-5, atol=1e-6)


def test_matchtxt_matches_reference():
    z = np.load(os.path.join(G, "matching_ref_inputs.npz"))
    text = (orc.matchtxt_record("0.jpg", "1.jpg", z["m_a"]) + orc.matchtxt_record("0.jpg", "2.jpg", z["m_b"]) +
            orc.matchtxt_record("1.jpg", "2.jpg", z["m_c"]))
    assert text == open(os.path.join(G, "matching_ref.txt")).read()


def test_pair_list_matches_reference():
    ref = [ln.split() for ln in open(os.path.join(G, "pairs_ref.txt")).read().splitlines()]
    got = orc.image_pairs(20, 3, [7, 15])
    assert [[f"{a}.jpg", f"{b}.jpg"] for a, b in got] == ref
    # 1000 images: 999 + 998 + 997 + 993 + 985 pairs
    assert len(orc.image_pairs(1000, 3, [7, 15])) == 4972


def test_sift_file_layout(tmp_path):
    """3-feature file = 20 + 3*5*4 + 3*128 + 4 = 468 bytes, 'SIFT' 'V4.0' header, 0xff 'EOF' trailer
    (sizes and leading bytes observed when the reference's FeatureData::saveSIFTB2 was run)."""
    loc = np.arange(15, dtype=np.float32).reshape(3, 5)
    des = (np.arange(3 * 128) % 251).astype(np.uint8).reshape(3, 128)
    p = str(tmp_path / "a.sift")
    assert orc.sift_write(p, loc, des) == 468
    raw = open(p, "rb").read()
    assert len(raw) == 468
    assert raw[:8] == bytes([0x53, 0x49, 0x46, 0x54, 0x56, 0x34, 0x2E, 0x30])
    assert np.frombuffer(raw[8:20], np.int32).tolist() == [3, 5, 128]
    assert raw[-4:] == b"\xffEOF"
    assert raw[20:80] == loc.tobytes() and raw[80:464] == des.tobytes()
    l2, d2 = orc.sift_read(p)
    np.testing.assert_array_equal(l2, loc)
    np.testing.assert_array_equal(d2, des)


def test_mat_file_layout(tmp_path):
    """First .mat with one 3-match record = 20 + 5680 + 212 = 5912 bytes (observed from the
    reference's MatchFile::WritePMatch); struct sizes 20 / 24 / 184."""
    a, b = str(tmp_path / "0"), str(tmp_path / "1")
    assert orc.mat_write_pmatch(a, b, 10, 12, [1, 4, 7], [0, 5, 9]) == 0
    for f in (a + ".mat", b + ".mat"):
        assert os.path.getsize(f) == 5912
    raw = open(a + ".mat", "rb").read()
    hdr = np.frombuffer(raw[:20], np.int32)
    assert raw[:4] == b"MT03" and hdr[1] == 1 and hdr[2] == 24 and hdr[3] == 5680 and hdr[4] == 0
    rec = np.frombuffer(raw[20:40], np.int32)
    assert rec.tolist() == [12, 5700, 184 + 8 * 24, 0, 0] and raw[40:42] == b"1\x00"
    assert np.frombuffer(raw[5700:5704], np.int32)[0] == 3 and raw[5704:5708] == b"MRV3"
    assert np.frombuffer(raw[5700 + 188:5700 + 212], np.int32).tolist() == [1, 4, 7, 0, 5, 9]
    ia, ib = orc.mat_get_pmatch(a, b, 12)
    assert ia.tolist() == [1, 4, 7] and ib.tolist() == [0, 5, 9]
    ib2, ia2 = orc.mat_get_pmatch(b, a, 10)   # the partner's file stores the rows swapped
    assert ia2.tolist() == [1, 4, 7] and ib2.tolist() == [0, 5, 9]
    assert len(orc.mat_get_pmatch(a, b, 11)[0]) == 0   # feature-count mismatch is rejected


def test_knn_u8_against_numpy():
    a, b = synth.sift_pair(120, 333, seed=3)
    b[40] = b[7]
    a[5] = a[90]
    D = ((a[:, None, :].astype(np.int64) - b[None].astype(np.int64)) ** 2).sum(-1)
    i, d1, d2 = orc.knn2_u8(a, b)
    np.testing.assert_array_equal(i, D.argmin(1))
    np.testing.assert_array_equal(d1, D.min(1))
    np.testing.assert_array_equal(d2, np.sort(D, 1)[:, 1])
    ci, cd = orc.colbest_u8(a, b)
    np.testing.assert_array_equal(ci, D.argmin(0))
    p = orc.params((4, 5), True)
    i0, i1, dd = orc.match_u8(a, b, p)
    keep = [(r, c) for r, c in enumerate(D.argmin(1))
            if D[r, c] * 25 < 16 * np.sort(D[r])[1] and D.argmin(0)[c] == r]
    assert list(zip(i0.tolist(), i1.tolist())) == keep
    assert len(keep) > 10


def test_knn_u8_degenerate():
    a = np.zeros((0, 128), np.uint8)
    b = synth.sift_like(4, np.random.default_rng(0))
    assert len(orc.match_u8(a, b, orc.params())[0]) == 0
    assert len(orc.match_u8(b, a, orc.params())[0]) == 0
    i, d1, d2 = orc.knn2_u8(b, b[:1])
    assert (i == 0).all() and (d2 == orc.NO_SECOND).all()
    # a single train row has no runner-up: the ratio test passes
    i0, i1, _ = orc.match_u8(b[:1], b[:1], orc.params((4, 5), True))
    assert i0.tolist() == [0] and i1.tolist() == [0]


def test_knn_f32_against_float64():
    a, b = synth.superpoint_pair(90, 140, 256, seed=4)
    S = a.astype(np.float64) @ b.astype(np.float64).T
    i, s1, s2 = orc.knn2_f32(a, b, orc.METRIC_DOT)
    np.testing.assert_array_equal(i, S.argmax(1))
    np.testing.assert_allclose(s1, S.max(1), rtol=1e-5, atol=1e-6)
    rng = np.random.default_rng(44)
    for r, eps in zip(range(0, 40, 4), np.logspace(-5, -1, 10)):      # near-duplicates at 1e-5 .. 1e-1
        b[r + 1] = a[r] + (eps / 16.0 * rng.standard_normal(256)).astype(np.float32)
    D = ((a[:, None, :].astype(np.float64) - b[None].astype(np.float64)) ** 2).sum(-1)
    i, d1, d2 = orc.knn2_f32(a, b, orc.METRIC_L2)
    np.testing.assert_array_equal(i, D.argmin(1))
    # difference-form distances: relative tolerance only, near-duplicates included
    np.testing.assert_allclose(d1, D.min(1), rtol=1e-5, atol=0)
    np.testing.assert_allclose(d2, np.sort(D, 1)[:, 1], rtol=1e-5, atol=0)
    assert D.min() < 1e-9          # the planted near-duplicates are part of the comparison


def test_nearest_embed_near_ties_fixture():
    """Codes in pairs 1e-6 apart and points 1e-4 .. 1e-2 from them (made by calling the reference's
    nearest_embed, tests/golden/gen_golden.py): the oracle finds the reference's argmin everywhere, which a
    distance of the form |x|^2 + |e|^2 - 2 x.e cannot (its rounding error exceeds the gaps)."""
    z = np.load(os.path.join(G, "nearest_embed_near_ties.npz"))
    res, am = orc.nearest_embed_f32(z["x"], z["emb"])
    np.testing.assert_array_equal(am, z["argmin"])
    np.testing.assert_array_equal(res, z["result"])
