"""Host-side product code (file formats, pair lists, C-ABI surface) — no GPU needed.

The product's writers/readers are checked against the reference-generated fixtures and,
byte for byte, against the oracle's independent restatement."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import _binding as B
from visual_slam_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


# ---- C ABI surface ----------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vsm.h")).read()
    declared = set(re.findall(r"\b(vsm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    lib = C.CDLL(vsm.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), "libvsm.so does not export %s" % name
    assert declared == set(B.SIGNATURES), "binding table and header disagree: %s" % (declared ^ set(B.SIGNATURES))


def test_create_fails_loudly_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(vsm.VsmError) as e:
        vsm.Matcher(0)
    assert "HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_merge_top2_host():
    rng = np.random.default_rng(1)
    ns, nq = 4, 50
    k1 = rng.random((ns, nq)).astype(np.float32)
    k2 = k1 + rng.random((ns, nq)).astype(np.float32)
    i1 = rng.integers(0, 100, (ns, nq)).astype(np.int32)
    i2 = rng.integers(0, 100, (ns, nq)).astype(np.int32)
    k1[0, 5] = k1[2, 5] = -1.0   # tie between slices at the top: the lower global index wins
    base = np.array([0, 100, 200, 300, 400], np.int64)
    ok1, ok2, oi1, oi2 = B.merge_top2_f32(k1, k2, i1, i2, base, vsm.METRIC_L2)
    gi1 = i1 + base[:ns, None].astype(np.int32)
    gi2 = i2 + base[:ns, None].astype(np.int32)
    e1, e2, ei1, ei2 = orc.merge_top2_f32(k1, k2, gi1, gi2, orc.METRIC_L2)
    np.testing.assert_array_equal(ok1, e1)
    np.testing.assert_array_equal(ok2, e2)
    np.testing.assert_array_equal(oi1, ei1)
    np.testing.assert_array_equal(oi2, ei2)
    assert oi1[5] == gi1[0, 5]


# ---- .sift ----------------------------------------------------------------------------------------
def test_sift_roundtrip_and_bytes(tmp_path):
    rng = np.random.default_rng(2)
    for n in (1, 3, 777):
        loc = rng.random((n, 5)).astype(np.float32) * 100
        des = rng.integers(0, 256, (n, 128), dtype=np.uint8)
        p, q = str(tmp_path / ("p%d.sift" % n)), str(tmp_path / ("o%d.sift" % n))
        host.write_sift(p, loc, des)
        orc.sift_write(q, loc, des)
        assert open(p, "rb").read() == open(q, "rb").read()
        assert os.path.getsize(p) == 20 + n * 20 + n * 128 + 4
        l2, d2 = host.read_sift(p)
        np.testing.assert_array_equal(l2, loc)
        np.testing.assert_array_equal(d2, des)
        l3, d3 = host.read_sift(p, nmax=max(1, n // 2))
        np.testing.assert_array_equal(d3, des[:max(1, n // 2)])
        np.testing.assert_array_equal(l3, loc[:max(1, n // 2)])


def test_sift_read_returns_what_the_reference_call_returns(tmp_path):
    """FeatureData::ReadSIFTB(file, locbuf, desbuf, nmax) returns min(nmax, npoint), the rows now in the caller's
    buffers (FeaturePoints.cpp:524-529): a caller written against it may use the value as exactly that."""
    import ctypes as C
    from visual_slam_amd import _binding as B
    rng = np.random.default_rng(3)
    loc = rng.random((10, 5)).astype(np.float32)
    des = rng.integers(0, 256, (10, 128), dtype=np.uint8)
    p = str(tmp_path / "ten.sift").encode()
    host.write_sift(p.decode(), loc, des)
    lib = B.lib()
    npoint, nloc = C.c_int(0), C.c_int(0)
    assert lib.vsm_sift_read(p, None, None, 0, C.byref(npoint), C.byref(nloc)) == 10      # header query
    assert (npoint.value, nloc.value) == (10, 5)
    for nmax, want in ((4, 4), (10, 10), (25, 10), (0, 0), (-3, 0)):
        lb = np.full((25, 5), -1.0, np.float32)
        db = np.full((25, 128), 7, np.uint8)
        assert lib.vsm_sift_read(p, B._ptr(lb), B._ptr(db), nmax, C.byref(npoint), None) == want
        assert npoint.value == 10
        np.testing.assert_array_equal(db[:want], des[:want])
        np.testing.assert_array_equal(lb[:want], loc[:want])
        assert (db[want:] == 7).all() and (lb[want:] == -1.0).all()                       # nothing past the rows returned
    assert lib.vsm_sift_read(p, None, B._ptr(db), 3, None, None) == 3                     # descriptors only


def test_sift_reader_rejects_bad_files(tmp_path):
    p = str(tmp_path / "bad.sift")
    loc = np.zeros((2, 5), np.float32)
    des = np.ones((2, 128), np.uint8)
    host.write_sift(p, loc, des)
    raw = bytearray(open(p, "rb").read())
    for off, val in ((0, b"XIFT"), (4, b"V9.0"), (16, np.int32(64).tobytes()), (8, np.int32(0).tobytes())):
        bad = bytearray(raw)
        bad[off:off + 4] = val
        open(p, "wb").write(bad)
        with pytest.raises(vsm.VsmError):
            host.read_sift(p)
    with pytest.raises(vsm.VsmError):
        host.read_sift(str(tmp_path / "missing.sift"))
    # the reference's txt->sift path writes descriptors of all ones (FeaturePoints.cpp:358-367)
    open(p, "wb").write(raw)
    _, d = host.read_sift(p)
    assert (d == 1).all()


# ---- match txt / keypoint txt / pair list -------------------------------------------------------
def test_matchtxt_matches_reference(tmp_path):
    z = np.load(os.path.join(G, "matching_ref_inputs.npz"))
    ref = open(os.path.join(G, "matching_ref.txt")).read()
    p = str(tmp_path / "matching.txt")
    assert host.save_matching_sfm(p, z["m_a"], "0.jpg", "1.jpg") == 3
    assert host.save_matching_sfm(p, z["m_b"], "0.jpg", "2.jpg") == 4
    assert host.save_matching_sfm(p, z["m_c"], "1.jpg", "2.jpg") == 0
    assert open(p).read() == ref
    # list form
    q = str(tmp_path / "matching2.txt")
    for m, a, b in ((z["m_a"], "0.jpg", "1.jpg"), (z["m_b"], "0.jpg", "2.jpg"), (z["m_c"], "1.jpg", "2.jpg")):
        i0 = np.nonzero(m != -1)[0]
        host.save_matching_list(q, i0, m[i0], len(m), a, b)
    assert open(q).read() == ref
    # python mirror of the string builder
    l0, l1, n = host.get_list_matching_pairs(z["m_a"])
    assert (l0, l1, n) == ("1 4 11", "7 0 3", 3)
    l0, l1, n = host.get_list_matching_pairs(z["m_b"])
    assert (l0, l1, n) == ("0 2 3 7 ", "5 8 1 2 ", 4)


def test_feature_txt_matches_reference(tmp_path):
    kp = np.load(os.path.join(G, "feature_ref_inputs.npy"))
    p = str(tmp_path / "0.txt")
    host.save_feature_locations(kp, p)
    assert open(p).read() == open(os.path.join(G, "feature_ref.txt")).read()
    back = host.read_feature_locations(p)
    np.testing.assert_array_equal(back, np.trunc(kp))


def test_pair_lists():
    names = [str(i) + ".jpg" for i in range(20)]
    ref = [ln.split() for ln in open(os.path.join(G, "pairs_ref.txt")).read().splitlines()]
    assert host.generate_imagePairs(names) == ref
    ex = host.generate_exhaustive_pairs(7)
    assert len(ex) == 21 and ex[0].tolist() == [0, 1] and ex[-1].tolist() == [5, 6]
    assert len(host.generate_exhaustive_pairs(1000)) == 499500


# ---- .mat ---------------------------------------------------------------------------------------------
def test_mat_bytes_match_oracle(tmp_path):
    """A sequence of writes, including re-writes of an existing partner, gives the same bytes from
    the product writer and from the oracle's restatement, and each reads the other's files."""
    rng = np.random.default_rng(3)
    d1, d2 = tmp_path / "p", tmp_path / "o"
    d1.mkdir()
    d2.mkdir()
    fc = [50, 60, 70, 80]
    seq = [(0, 1, 5), (0, 2, 9), (1, 2, 0), (0, 3, 20), (0, 3, 12), (2, 3, 33)]
    for a, b, nm in seq:
        ia = np.sort(rng.choice(fc[a], nm, replace=False)).astype(np.int32)
        ib = rng.integers(0, fc[b], nm).astype(np.int32)
        host.write_pmatch(str(d1 / str(a)), str(d1 / str(b)), fc[a], fc[b], ia, ib)
        assert orc.mat_write_pmatch(str(d2 / str(a)), str(d2 / str(b)), fc[a], fc[b], ia, ib) == 0
        last = (a, b, ia, ib)
    for i in range(4):
        assert open(d1 / ("%d.mat" % i), "rb").read() == open(d2 / ("%d.mat" % i), "rb").read()
    a, b, ia, ib = last
    ga, gb = host.get_pmatch(str(d2 / str(a)), str(d2 / str(b)), fc[b])      # product reads oracle's file
    np.testing.assert_array_equal(ga, ia)
    np.testing.assert_array_equal(gb, ib)
    oa, ob = orc.mat_get_pmatch(str(d1 / str(b)), str(d1 / str(a)), fc[a])     # oracle reads product's file
    np.testing.assert_array_equal(oa, ib)
    np.testing.assert_array_equal(ob, ia)
    assert host.get_match_count(str(d1 / str(a)), str(d1 / str(b))) == (33, 0)
    assert host.get_match_count(str(d1 / "0"), str(d1 / "9")) is None
    ra, rb = host.get_pmatch_r(str(d1 / str(a)), str(d1 / str(b)), fc[b])   # rows exchanged
    pa, pb = host.get_pmatch(str(d1 / str(a)), str(d1 / str(b)), fc[b])
    np.testing.assert_array_equal(ra, pb)
    np.testing.assert_array_equal(rb, pa)
    assert len(host.get_pmatch(str(d1 / str(a)), str(d1 / str(b)), fc[b] + 1)[0]) == 0


def test_mat_outgrown_blocks_and_full_directory_match_oracle(tmp_path):
    """Re-writes that outgrow a block in the middle of a file and the block of record 0 (the record moves to the
    end: MatchFile::MoveRecordToEnd, MatchFile.cpp:422-453, with its bookkeeping of trash_size and
    definition_buf), a re-write that fits block + trash, and more partners than the directory holds
    (AddImageMatch's regrow, :494-560): product and oracle leave the same bytes, and every record reads back."""
    rng = np.random.default_rng(31)
    d1, d2 = tmp_path / "p", tmp_path / "o"
    d1.mkdir()
    d2.mkdir()
    n = 8
    fc = [300 + 10 * k for k in range(n)]
    truth = {}

    def put(a, b, nm):
        ia = np.sort(rng.choice(fc[a], nm, replace=False)).astype(np.int32)
        ib = rng.integers(0, fc[b], nm).astype(np.int32)
        host.write_pmatch(str(d1 / ("i%d" % a)), str(d1 / ("i%d" % b)), fc[a], fc[b], ia, ib)
        assert orc.mat_write_pmatch(str(d2 / ("i%d" % a)), str(d2 / ("i%d" % b)), fc[a], fc[b], ia, ib) == 0
        truth[(a, b)] = (ia, ib)

    for b in range(1, n):
        put(0, b, 10 + b)                 # file i0: records for i1 .. i7, in that order
    put(0, 3, 200)                        # outgrows a block in the middle: record i3 moves behind i7
    put(0, 1, 250)                        # outgrows the block of record 0
    put(0, 5, 12)                         # smaller: stays where it is
    put(0, 4, 40)                         # needs more than its block, fits block + trash? (else it moves)
    put(0, 2, 300)
    put(0, 3, 260)                        # the moved record again, now not the last one
    for a in range(n):
        x = (d1 / ("i%d.mat" % a)).read_bytes()
        y = (d2 / ("i%d.mat" % a)).read_bytes()
        assert x == y, "i%d.mat differs" % a
    for (a, b), (ia, ib) in truth.items():
        ga, gb = host.get_pmatch(str(d1 / ("i%d" % a)), str(d1 / ("i%d" % b)), fc[b])
        np.testing.assert_array_equal(ga, ia)
        np.testing.assert_array_equal(gb, ib)
        oa, ob = orc.mat_get_pmatch(str(d2 / ("i%d" % b)), str(d2 / ("i%d" % a)), fc[a])
        np.testing.assert_array_equal(oa, ib)
        np.testing.assert_array_equal(ob, ia)
    hdr = np.frombuffer((d1 / "i0.mat").read_bytes()[:20], np.int32)
    assert hdr[3] > 5680                  # definition_buf grew by the length of record 0's directory entry

    # a full directory: 300 partners with a reservation of 20 records (5680 bytes)
    hub1, hub2 = str(tmp_path / "p" / "hub"), str(tmp_path / "o" / "hub")
    lists = []
    for k in range(300):
        nm = int(rng.integers(1, 30))
        ia = np.sort(rng.choice(100, nm, replace=False)).astype(np.int32)
        ib = rng.integers(0, 40 + k, nm).astype(np.int32)
        lists.append((ia, ib))
        host.write_pmatch(hub1, str(tmp_path / "p" / ("img%03d" % k)), 100, 40 + k, ia, ib)
        assert orc.mat_write_pmatch(hub2, str(tmp_path / "o" / ("img%03d" % k)), 100, 40 + k, ia, ib) == 0
    # and blocks outgrown after the directory has been rotated
    for k in (0, 150, 7):
        ia = np.arange(40, dtype=np.int32)
        ib = np.arange(40, dtype=np.int32)[::-1].copy()
        lists[k] = (ia, ib)
        host.write_pmatch(hub1, str(tmp_path / "p" / ("img%03d" % k)), 100, 40 + k, ia, ib)
        assert orc.mat_write_pmatch(hub2, str(tmp_path / "o" / ("img%03d" % k)), 100, 40 + k, ia, ib) == 0
    assert open(hub1 + ".mat", "rb").read() == open(hub2 + ".mat", "rb").read()
    for k in range(300):
        ga, gb = host.get_pmatch(hub1, str(tmp_path / "p" / ("img%03d" % k)), 40 + k)
        np.testing.assert_array_equal(ga, lists[k][0])
        np.testing.assert_array_equal(gb, lists[k][1])


def test_mat_first_file_size(tmp_path):
    a, b = str(tmp_path / "0"), str(tmp_path / "1")
    host.write_pmatch(a, b, 10, 12, [1, 4, 7], [0, 5, 9])
    assert os.path.getsize(a + ".mat") == 5912 and os.path.getsize(b + ".mat") == 5912


def test_mat_directory_growth(tmp_path):
    """More partners than the directory was reserved for: the directory doubles and every record
    stays readable (the reference's AddImageMatch regrow path, MatchFile.cpp:494-560)."""
    rng = np.random.default_rng(4)
    n = 300
    truth = {}
    hub = str(tmp_path / "hub")
    for k in range(n):
        nm = int(rng.integers(1, 30))
        ia = np.sort(rng.choice(100, nm, replace=False)).astype(np.int32)
        ib = rng.integers(0, 40 + k, nm).astype(np.int32)
        host.write_pmatch(hub, str(tmp_path / ("img%03d" % k)), 100, 40 + k, ia, ib)
        truth[k] = (ia, ib)
    for k in range(n):
        ga, gb = host.get_pmatch(hub, str(tmp_path / ("img%03d" % k)), 40 + k)
        np.testing.assert_array_equal(ga, truth[k][0])
        np.testing.assert_array_equal(gb, truth[k][1])
        # the oracle's reader walks the same directory
        oa, ob = orc.mat_get_pmatch(hub, str(tmp_path / ("img%03d" % k)), 40 + k)
        np.testing.assert_array_equal(oa, truth[k][0])
        np.testing.assert_array_equal(ob, truth[k][1])
    # with a reservation the directory never has to grow
    hub2 = str(tmp_path / "hub2")
    for k in range(n):
        host.write_pmatch(hub2, str(tmp_path / ("jmg%03d" % k)), 100, 40 + k, truth[k][0], truth[k][1], reservation=n)
    assert np.frombuffer(open(hub2 + ".mat", "rb").read(20), np.int32)[3] == n * 284
    ga, gb = host.get_pmatch(hub2, str(tmp_path / "jmg299"), 40 + 299)
    np.testing.assert_array_equal(ga, truth[299][0])


def test_ratio_fraction():
    assert B.ratio_to_fraction(0.8) == (4, 5)
    assert B.ratio_to_fraction(0.75) == (3, 4)
    assert B.ratio_to_fraction(None) == (0, 1)
    n, d = B.ratio_to_fraction(0.7)
    assert (n, d) == (7, 10)


def test_mat_writer_leaves_the_same_bytes_as_repeated_calls(tmp_path):
    """Cached writer (few descriptors open, directories in memory) against one call per pair, with a
    small reservation (directory regrow) and pairs written twice with more matches (block moves)."""
    rng = np.random.default_rng(77)
    n = 30
    fc = [int(x) for x in rng.integers(50, 400, n)]
    pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    puts = []
    for a, b in pairs:
        nm = int(rng.integers(0, min(fc[a], fc[b])))
        puts.append((a, b, np.sort(rng.choice(fc[a], nm, replace=False)).astype(np.int32),
                     rng.choice(fc[b], nm, replace=False).astype(np.int32)))
    for a, b in pairs[::7]:     # the same pair again, with more matches than its block holds
        nm = min(fc[a], fc[b])
        puts.append((a, b, np.arange(nm, dtype=np.int32), rng.permutation(fc[b])[:nm].astype(np.int32)))
    d1, d2 = tmp_path / "calls", tmp_path / "writer"
    d1.mkdir()
    d2.mkdir()
    for a, b, i0, i1 in puts:
        host.write_pmatch(str(d1 / ("im%d" % a)), str(d1 / ("im%d" % b)), fc[a], fc[b], i0, i1, reservation=4)
    with host.MatWriter(reservation=4, max_open=5) as w:
        for a, b, i0, i1 in puts:
            w.put(str(d2 / ("im%d" % a)), str(d2 / ("im%d" % b)), fc[a], fc[b], i0, i1)
    for a in range(n):
        x = (d1 / ("im%d.mat" % a)).read_bytes()
        y = (d2 / ("im%d.mat" % a)).read_bytes()
        assert x == y, "im%d.mat differs" % a
    a, b, i0, i1 = puts[-1]
    ga, gb = host.get_pmatch(str(d2 / ("im%d" % a)), str(d2 / ("im%d" % b)), fc[b])
    np.testing.assert_array_equal(ga, i0)
    np.testing.assert_array_equal(gb, i1)


def test_resort_by_score_fixture():
    """sort_scores / sort_feature_Or_descriptors / get_correct_new_matching_index against the outputs of the
    reference's own functions (main.py:75-112, executed by tests/golden/gen_golden.py on scores with ties)."""
    z = np.load(os.path.join(G, "resort_ref.npz"))
    s0, i0 = host.sort_scores(z["scores0"])
    s1, i1 = host.sort_scores(z["scores1"])
    np.testing.assert_array_equal(s0, z["n_scores0"])
    np.testing.assert_array_equal(i0, z["idx0"])
    np.testing.assert_array_equal(s1, z["n_scores1"])
    np.testing.assert_array_equal(i1, z["idx1"])
    for got, want in ((host.sort_feature_Or_descriptors(z["kpts0"], i0), z["n_kpts0"]),
                      (host.sort_feature_Or_descriptors(z["kpts1"], i1), z["n_kpts1"]),
                      (host.sort_feature_Or_descriptors(z["conf"], i0), z["n_conf"])):
        assert got.dtype == want.dtype
        np.testing.assert_array_equal(got, want)
    nm = host.get_correct_new_matching_index(host.sort_feature_Or_descriptors(z["matches"], i0), i1)
    assert nm.dtype == z["n_matches"].dtype
    np.testing.assert_array_equal(nm, z["n_matches"])
    k0, k1, nm2 = host.resort_pair_by_score(z["kpts0"], z["scores0"], z["kpts1"], z["scores1"], z["matches"])
    np.testing.assert_array_equal(k0, z["n_kpts0"])
    np.testing.assert_array_equal(k1, z["n_kpts1"])
    np.testing.assert_array_equal(nm2, z["n_matches"])
    # and the list the exporter writes from it
    l0, l1, n = host.get_list_matching_pairs(nm2)
    assert n == 20 and len(l0.split()) == 20 and len(l1.split()) == 20


def test_inlier_records_that_outgrow_their_block_match_oracle(tmp_path):
    """WriteIMatch's re-allocation branch (MatchFile.cpp:690-719): inlier lists too long for a record's block, for the
    last record of a file (larger block in place), a record in the middle and record 0 (both move to the end, NM and
    the putative lists with them), and one that fits its block plus the dead bytes behind it.  Product and oracle
    leave the same bytes in every file, the cached writer too, and every list reads back."""
    rng = np.random.default_rng(77)
    dirs = [tmp_path / name for name in ("p", "o", "w")]
    for d in dirs:
        d.mkdir()
    n, fc = 6, 500
    F = rng.standard_normal((3, 3)).astype(np.float32)
    put, inl = {}, {}
    steps = []
    for b in range(1, n):
        nm = 8 if b == 1 else 2 + b
        steps.append(("p", 0, b, np.sort(rng.choice(fc, nm, replace=False)).astype(np.int32), rng.integers(0, fc, nm).astype(np.int32)))
    for b, nf in ((5, 40), (2, 45), (3, 3), (1, 16), (2, 70), (4, 0)):
        # i5: the last record; i2: the middle, moves (its predecessor i1 gains dead bytes); i3: fits; i1 (NM = 8): fits
        # its block plus those dead bytes; i2 again, now the last record; i4: no inliers at all
        steps.append(("f", 0, b, np.sort(rng.choice(fc, nf, replace=False)).astype(np.int32), rng.integers(0, fc, nf).astype(np.int32)))
    steps.append(("f", 1, 0, np.arange(60, dtype=np.int32), np.arange(60, dtype=np.int32)[::-1].copy()))   # record 0 of file i1 moves
    import ctypes as C
    from visual_slam_amd import _binding as B
    lib = B.lib()
    w = lib.vsm_mat_writer_open(0, 2)
    for kind, a, b, i0, i1 in steps:
        pa, pb = ["i%d" % a, "i%d" % b]
        if kind == "p":
            host.write_pmatch(str(dirs[0] / pa), str(dirs[0] / pb), fc, fc, i0, i1)
            assert orc.mat_write_pmatch(str(dirs[1] / pa), str(dirs[1] / pb), fc, fc, i0, i1) == 0
            assert lib.vsm_mat_writer_put(w, str(dirs[2] / pa).encode(), str(dirs[2] / pb).encode(), fc, fc, len(i0), B._ptr(i0), B._ptr(i1)) == 0
            put[(a, b)] = (i0, i1)
        else:
            assert host.write_fmatch(str(dirs[0] / pa), str(dirs[0] / pb), i0, i1, F) == 1
            assert orc.mat_write_fmatch(str(dirs[1] / pa), str(dirs[1] / pb), i0, i1, F) == 1
            Ff = np.ascontiguousarray(F.reshape(9))
            assert lib.vsm_mat_writer_put_fmatch(w, str(dirs[2] / pa).encode(), str(dirs[2] / pb).encode(), len(i0), B._ptr(i0), B._ptr(i1), B._ptr(Ff)) == 1
            inl[(min(a, b), max(a, b))] = (i0, i1, F) if a < b else (i1, i0, F.T)   # the first image's file holds F itself
    assert lib.vsm_mat_writer_close(w) == 0
    for k in range(n):
        x, y, z = [(d / ("i%d.mat" % k)).read_bytes() for d in dirs]
        assert x == y, "i%d.mat: product and oracle differ" % k
        assert x == z, "i%d.mat: the cached writer differs" % k
    for (a, b), (i0, i1) in put.items():
        ga, gb = host.get_pmatch(str(dirs[0] / ("i%d" % a)), str(dirs[0] / ("i%d" % b)), fc)
        np.testing.assert_array_equal(ga, i0)         # the putative lists moved with their record
        np.testing.assert_array_equal(gb, i1)
        gb2, ga2 = host.get_pmatch(str(dirs[0] / ("i%d" % b)), str(dirs[0] / ("i%d" % a)), fc)
        np.testing.assert_array_equal(ga2, i0)
        np.testing.assert_array_equal(gb2, i1)
    for (a, b), (i0, i1, Fab) in inl.items():
        nm, Fm, r0, r1 = host.get_imatch(str(dirs[0] / ("i%d" % a)), str(dirs[0] / ("i%d" % b)), fc)
        assert nm == len(put[(a, b)][0])
        np.testing.assert_array_equal(r0, i0)
        np.testing.assert_array_equal(r1, i1)
        if len(i0):
            np.testing.assert_array_equal(Fm, Fab)
        nm, Fm, r1b, r0b = host.get_imatch(str(dirs[0] / ("i%d" % b)), str(dirs[0] / ("i%d" % a)), fc)
        np.testing.assert_array_equal(r0b, i0)
        np.testing.assert_array_equal(r1b, i1)
        if len(i0):
            np.testing.assert_array_equal(Fm, Fab.T)
    hdr = np.frombuffer((dirs[0] / "i1.mat").read_bytes()[:20], np.int32)
    assert hdr[1] == 1 or hdr[3] >= 5680


def _mat_writer_process(args):
    """One of several processes that write putative and inlier records of pairs (0, k) into the same i0.mat."""
    folder, ks, lock, start_at = args
    import time
    import numpy as np
    from visual_slam_amd import _binding as B
    from visual_slam_amd import host
    B.lib().vsm_mat_set_locking(1 if lock else 0)
    F = np.eye(3, dtype=np.float32)
    while time.time() < start_at:      # all writers start together
        pass
    for k in ks:
        i0 = np.arange(20 + k % 7, dtype=np.int32)
        i1 = (i0 * 3 + k) % 1000
        host.write_pmatch("%s/i0" % folder, "%s/i%d" % (folder, k), 1000, 1000, i0, i1.astype(np.int32))
        host.write_fmatch("%s/i0" % folder, "%s/i%d" % (folder, k), i0[:5], i1[:5].astype(np.int32), F)
    return len(ks)


def test_processes_share_a_mat_file_under_locking(tmp_path):
    """vsm_mat_set_locking (the counterpart of MatchFile::multi_thread_mode, MatchFile.cpp:221-231): four processes
    write disjoint pairs that all involve image 0, so every one of them opens, extends and re-writes the directory of
    i0.mat (which also fills up and doubles on the way); with the lock held from open to close no record is lost."""
    import multiprocessing as mp
    import time
    nproc, per = 4, 150
    start_at = time.time() + 3.0
    jobs = [(str(tmp_path), list(range(1 + p, 1 + nproc * per, nproc)), True, start_at) for p in range(nproc)]
    with mp.get_context("spawn").Pool(nproc) as pool:
        assert sum(pool.map(_mat_writer_process, jobs)) == nproc * per
    for k in range(1, 1 + nproc * per):
        i0 = np.arange(20 + k % 7, dtype=np.int32)
        i1 = ((i0 * 3 + k) % 1000).astype(np.int32)
        ga, gb = host.get_pmatch(str(tmp_path / "i0"), str(tmp_path / ("i%d" % k)), 1000)
        np.testing.assert_array_equal(ga, i0)
        np.testing.assert_array_equal(gb, i1)
        nm, Fm, r0, r1 = host.get_imatch(str(tmp_path / "i0"), str(tmp_path / ("i%d" % k)), 1000)
        assert nm == len(i0)
        np.testing.assert_array_equal(r0, i0[:5])
        np.testing.assert_array_equal(r1, i1[:5])
        gb2, ga2 = host.get_pmatch(str(tmp_path / ("i%d" % k)), str(tmp_path / "i0"), 1000)
        np.testing.assert_array_equal(ga2, i0)
