"""GPU parity: SIFT-128 u8 matcher (HIP, through the C ABI) against the CPU oracle.

Bar: bit-exact indices, distances and match lists (integer arithmetic)."""
import numpy as np
import pytest

import oracle_lib as orc
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1, 3], ids=["tilemin", "packed", "tilemin-fused"])
def u8_mode(matcher, request):
    """Every test runs three times: with the tile-minimum query pass for ratio-test submissions (default: candidates
    settled in the query kernel's tail for lists of eight pairs or more), with the packed form only, and with the
    tail form forced for every submission whose sizes allow it.  Same bits are demanded of all."""
    matcher.set_u8_mode(request.param)
    yield request.param
    matcher.set_u8_mode(0)


SIZES = [(1, 1), (1, 5), (5, 1), (2, 2), (37, 91), (256, 256), (257, 255), (300, 700), (777, 2048), (1000, 513)]


def _check_pair(matcher, a, b, ratio=(4, 5)):
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        gi, gd1, gd2 = matcher.knn2(s0, s1)
        oi, od1, od2 = orc.knn2_u8(a, b)
        np.testing.assert_array_equal(gd1, od1)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd2, od2)
        for mutual in (True, False):
            fr = ratio[0] / ratio[1] if ratio else None
            g0, g1, gd = matcher.match_pair(s0, s1, ratio=fr, mutual=mutual, return_dist=True)
            o0, o1, od = orc.match_u8(a, b, orc.params(ratio, mutual))
            np.testing.assert_array_equal(g0, o0)
            np.testing.assert_array_equal(g1, o1)
            np.testing.assert_array_equal(gd, od)
            gm0, gm1 = matcher.match_pair_dense(s0, s1, ratio=fr, mutual=mutual)
            om0, om1 = orc.match_dense_u8(a, b, orc.params(ratio, mutual))
            np.testing.assert_array_equal(gm0, om0)
            np.testing.assert_array_equal(gm1, om1)
        return len(o0)
    finally:
        matcher.release(s0)
        matcher.release(s1)


@pytest.mark.parametrize("n0,n1", SIZES)
def test_pair_sizes(matcher, n0, n1):
    a, b = synth.sift_pair(n0, n1, seed=1000 + n0 * 7 + n1)
    _check_pair(matcher, a, b)


def test_planted_matches_survive(matcher):
    a, b = synth.sift_pair(777, 2048, seed=5)
    nm = _check_pair(matcher, a, b)
    assert nm > 100  # the planted correspondences pass ratio + mutual


def test_no_ratio(matcher):
    a, b = synth.sift_pair(300, 400, seed=6)
    _check_pair(matcher, a, b, ratio=None)


def test_ties_lowest_index(matcher):
    """Duplicate rows on both sides: every tie must resolve to the lowest index."""
    rng = np.random.default_rng(7)
    base = synth.sift_like(40, rng)
    a = np.concatenate([base, base[:17], base[5:9]])          # duplicate queries
    b = np.concatenate([base[::-1], base[3:30], base[:40]])     # duplicate train rows
    _check_pair(matcher, a, b)
    _check_pair(matcher, a, b, ratio=None)


def test_extreme_values(matcher):
    """All-0 and all-255 rows reach the largest possible distance 128*255^2."""
    a = np.zeros((70, 128), np.uint8)
    b = np.full((90, 128), 255, np.uint8)
    a[1::2] = 255
    b[::3] = 0
    rng = np.random.default_rng(8)
    a[10:20] = rng.integers(0, 256, (10, 128), dtype=np.uint8)
    b[40:60] = rng.integers(0, 256, (20, 128), dtype=np.uint8)
    _check_pair(matcher, a, b, ratio=None)
    _check_pair(matcher, a, b)


def test_extreme_values_over_a_whole_chunk(matcher):
    """The query pass ranks half-distance * 256 + (tile within the chunk): the largest and the smallest
    half-distance there is, in every one of the 256 tiles of an 8192-row chunk, with the nearest row of some
    queries in the last tile."""
    rng = np.random.default_rng(18)
    b = np.full((8192, 128), 255, np.uint8)
    b[1::2] = 0
    b[8190] = rng.integers(0, 256, 128, dtype=np.uint8)
    a = np.zeros((96, 128), np.uint8)          # all-0 queries: distance 0 to the odd rows, the maximum to the even ones
    a[32:64] = 255                             # all-255 queries: the other way round
    a[64:] = rng.integers(0, 256, (32, 128), dtype=np.uint8)
    a[70] = b[8190]                            # nearest row in the last tile
    a[71] = np.clip(b[8190].astype(np.int32) + rng.integers(-2, 3, 128), 0, 255).astype(np.uint8)
    _check_pair(matcher, a, b)
    _check_pair(matcher, a, b, ratio=None)
    _check_pair(matcher, b[:700], a)


def test_uniform_random_bytes(matcher):
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (513, 128), dtype=np.uint8)
    b = rng.integers(0, 256, (700, 128), dtype=np.uint8)
    _check_pair(matcher, a, b)


def test_empty_sets(matcher):
    rng = np.random.default_rng(10)
    a = synth.sift_like(10, rng)
    e = np.zeros((0, 128), np.uint8)
    s0, s1, se = matcher.upload_u8(a), matcher.upload_u8(a), matcher.upload_u8(e)
    for x, y in ((s0, se), (se, s1), (se, se)):
        i0, i1 = matcher.match_pair(x, y)
        assert len(i0) == 0 and len(i1) == 0
    off, i0, i1 = matcher.match_pairs([[s0, se], [s0, s1], [se, s1]])
    assert off[0] == 0 and off[1] == 0 and off[3] == off[2]
    o0, o1, _ = orc.match_u8(a, a, orc.params((4, 5), True))
    np.testing.assert_array_equal(i0, o0)
    np.testing.assert_array_equal(i1, o1)
    for s in (s0, s1, se):
        matcher.release(s)


def test_train_set_larger_than_one_chunk(matcher):
    """More than 8192 train rows: per-chunk partial results are merged."""
    a, b = synth.sift_pair(300, 8192 + 300, seed=11)
    # plant exact duplicates across the chunk boundary to exercise the tie-break there
    b[8192 + 5] = b[100]
    b[8192 + 6] = a[7]
    b[50] = a[7]
    _check_pair(matcher, a, b)


def test_batch_of_pairs(matcher):
    rng = np.random.default_rng(12)
    sizes = [300, 0, 777, 64, 1025]
    imgs = []
    base = synth.sift_like(400, rng)
    for n in sizes:
        x = synth.sift_like(n, rng)
        k = min(n, 400) // 2
        if k:
            x[:k] = np.clip(base[:k].astype(np.int32) + rng.integers(-4, 5, (k, 128)), 0, 255).astype(np.uint8)
        imgs.append(x)
    ids = [matcher.upload_u8(x) for x in imgs]
    pairs = [(i, j) for i in range(len(ids)) for j in range(len(ids)) if i != j]
    off, i0, i1 = matcher.match_pairs([[ids[i], ids[j]] for i, j in pairs])
    assert off[0] == 0 and off[-1] == len(i0)
    for k, (i, j) in enumerate(pairs):
        o0, o1, _ = orc.match_u8(imgs[i], imgs[j], orc.params((4, 5), True))
        np.testing.assert_array_equal(i0[off[k]:off[k + 1]], o0)
        np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o1)
    for s in ids:
        matcher.release(s)


def test_many_submissions_in_flight(matcher):
    """A small scratch budget cuts the pair list into many submissions, which match_pairs keeps two at a
    time in flight: the lists must come back in pair order all the same."""
    rng = np.random.default_rng(13)
    sizes = [900, 513, 0, 1300, 64, 2100, 777, 1]
    base = synth.sift_like(500, rng)
    imgs = []
    for n in sizes:
        x = synth.sift_like(n, rng)
        k = min(n, 500) // 2
        if k:
            x[rng.choice(n, k, replace=False)] = np.clip(
                base[rng.choice(500, k, replace=False)].astype(np.int32) + rng.integers(-4, 5, (k, 128)), 0, 255).astype(np.uint8)
        imgs.append(x)
    ids = [matcher.upload_u8(x) for x in imgs]
    pairs = [(i, j) for i in range(len(ids)) for j in range(len(ids)) if i != j]
    matcher.set_scratch_budget(1 << 20)
    try:
        off, i0, i1 = matcher.match_pairs([[ids[i], ids[j]] for i, j in pairs])
    finally:
        matcher.set_scratch_budget(8 << 30)
    assert off[0] == 0 and off[-1] == len(i0)
    for k, (i, j) in enumerate(pairs):
        o0, o1, _ = orc.match_u8(imgs[i], imgs[j], orc.params((4, 5), True))
        np.testing.assert_array_equal(i0[off[k]:off[k + 1]], o0)
        np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o1)
    for s in ids:
        matcher.release(s)


def test_submission_too_large_to_bring_its_own_lists(matcher):
    """Small submissions copy offsets and lists to pinned host memory themselves; one with more than 262144 image-0 rows
    leaves them to copy commands on the second stream.  Both must hand back the same lists."""
    rng = np.random.default_rng(15)
    base = synth.sift_like(3000, rng)
    imgs = []
    for n in (7000, 7100, 6900, 7050):
        x = synth.sift_like(n, rng)
        rows = rng.choice(n, 2500, replace=False)
        x[rows] = np.clip(base[rng.choice(3000, 2500, replace=False)].astype(np.int32) + rng.integers(-3, 4, (2500, 128)), 0, 255).astype(np.uint8)
        imgs.append(x)
    ids = [matcher.upload_u8(x) for x in imgs]
    pairs = [(i, j) for i in range(4) for j in range(4) if i != j] * 4          # 48 pairs, 336 000 image-0 rows
    few = pairs[:12]                                                            # 84 000 rows: the pinned block
    try:
        off, i0, i1 = matcher.match_pairs([[ids[i], ids[j]] for i, j in pairs])
        soff, s0, s1 = matcher.match_pairs([[ids[i], ids[j]] for i, j in few])
        assert off[-1] == len(i0) and soff[-1] == len(s0) and off[-1] > 48 * 1000
        for k, (i, j) in enumerate(few):
            o0, o1, _ = orc.match_u8(imgs[i], imgs[j], orc.params((4, 5), True))
            for o, a0, a1 in ((off, i0, i1), (soff, s0, s1)):
                np.testing.assert_array_equal(a0[o[k]:o[k + 1]], o0)
                np.testing.assert_array_equal(a1[o[k]:o[k + 1]], o1)
        for k in range(12, 48):                                                 # the repeats of the same twelve pairs
            np.testing.assert_array_equal(i0[off[k]:off[k + 1]], i0[off[k % 12]:off[k % 12 + 1]])
            np.testing.assert_array_equal(i1[off[k]:off[k + 1]], i1[off[k % 12]:off[k % 12 + 1]])
    finally:
        for s in ids:
            matcher.release(s)


def test_full_size_pair_8192(matcher):
    """The benchmark's pair shape, checked element for element."""
    a, b = synth.sift_pair(8192, 8192, seed=14)
    assert _check_pair(matcher, a, b) > 500


def test_stress_pair_65536(matcher):
    """65536 x 65536 (eight 8192-row chunks per query strip), element for element against the oracle."""
    a, b = synth.sift_pair(65536, 65536, seed=15)
    b[60000] = a[123]      # duplicates far apart: ties across chunks
    b[70] = a[123]
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        gi, gd1, gd2 = matcher.knn2(s0, s1)
        oi, od1, od2 = orc.knn2_u8(a, b)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd1, od1)
        np.testing.assert_array_equal(gd2, od2)
        g0, g1 = matcher.match_pair(s0, s1, ratio=0.8, mutual=True)
        o0, o1, _ = orc.match_u8(a, b, orc.params((4, 5), True))
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
        assert gi[123] == 70 and len(o0) > 1000
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_submit_collect_keeps_order(matcher):
    """Two submissions in flight: lists come back oldest first; a third submit and an idle collect fail."""
    import visual_slam_amd as vsm
    rng = np.random.default_rng(16)
    imgs = [synth.sift_pair(700, 900, seed=100 + k) for k in range(3)]
    ids = [(matcher.upload_u8(a), matcher.upload_u8(b)) for a, b in imgs]
    batches = [[ids[0], ids[1]], [ids[2]], [ids[1], ids[0], ids[2]]]
    matcher.submit_pairs(batches[0])
    matcher.submit_pairs(batches[1])
    with pytest.raises(vsm.VsmError):
        matcher.submit_pairs(batches[2])
    got = [matcher.collect_matches()]
    matcher.submit_pairs(batches[2])
    got.append(matcher.collect_matches())
    got.append(matcher.collect_matches())
    with pytest.raises(vsm.VsmError):
        matcher._inflight.append((1, 1))
        matcher.collect_matches()
    order = [[0, 1], [2], [1, 0, 2]]
    for (off, i0, i1), which in zip(got, order):
        assert len(off) == len(which) + 1
        for k, w in enumerate(which):
            o0, o1, _ = orc.match_u8(imgs[w][0], imgs[w][1], orc.params((4, 5), True))
            np.testing.assert_array_equal(i0[off[k]:off[k + 1]], o0)
            np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o1)
    for a, b in ids:
        matcher.release(a)
        matcher.release(b)


def test_tile_minimum_path_is_the_one_that_runs(matcher, u8_mode):
    a, b = synth.sift_pair(600, 900, seed=17)
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        # 0: tile-minimum, 1: packed, 2: tile-minimum with the tail.  The default mode sends ONE pair of modest size
        # per call to the packed form (fewer launches), and lists to the tile-minimum form
        matcher.match_pair(s0, s1, ratio=0.8, mutual=True)
        assert matcher.last_u8_path() == {0: 1, 1: 1, 3: 2}[u8_mode]
        matcher.match_pairs([[s0, s1], [s1, s0]], ratio=0.8, mutual=True)
        assert matcher.last_u8_path() == {0: 0, 1: 1, 3: 2}[u8_mode]
        matcher.match_pairs([[s0, s1], [s1, s0]], ratio=None, mutual=True)
        assert matcher.last_u8_path() == {0: 0, 1: 1, 3: 2}[u8_mode]   # no ratio test: every row is resolved from its group
        matcher.knn2(s0, s1)
        assert matcher.last_u8_path() == 1                # raw 2-NN: the packed form in every mode
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_undecidable_candidates_are_rescanned(matcher):
    """Rows the one-tile resolve cannot settle: the nearest distance tied between rows of different tiles
    and of the two lane halves of one tile, runner-ups whose parity bit decides the ratio test, a runner-up
    inside the winner's 16-row group, every ratio from loose to strict."""
    rng = np.random.default_rng(18)
    b = synth.sift_like(3000, rng)
    a = synth.sift_like(400, rng)
    for k in range(100):                      # exact copies: d1 = 0, tied winners far apart and adjacent
        a[k] = b[7 * k]
    b[2000:2050] = b[0:350:7]                 # second copy in a later tile: tie on the minimum across tiles
    b[4:8] = b[0:4]                           # ... and across the lane halves of tile 0 (rows r and r + 4)
    b[2900] = b[0]
    for k in range(100, 200):                 # near copies with a runner-up one unit apart in one component
        src = b[9 * k % 3000].copy()
        a[k] = src
        t = src.copy()
        t[k % 128] = min(int(t[k % 128]) + 1, 255) if t[k % 128] < 255 else 254
        b[(9 * k % 3000) + 1] = t             # same tile, usually the same 16-row group or the other half
    for ratio in ((4, 5), (99, 100), (1, 2), (1, 100), (1, 1)):
        s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
        try:
            for mutual in (True, False):
                g0, g1, gd = matcher.match_pair(s0, s1, ratio=ratio[0] / ratio[1], mutual=mutual, return_dist=True)
                o0, o1, od = orc.match_u8(a, b, orc.params(ratio, mutual))
                np.testing.assert_array_equal(g0, o0)
                np.testing.assert_array_equal(g1, o1)
                np.testing.assert_array_equal(gd, od)
        finally:
            matcher.release(s0)
            matcher.release(s1)


def test_ties_between_chunks_with_merged_chunk_records(matcher):
    """A single pair of many query rows is cut into narrow train chunks whose partial results are merged by a
    pass of their own before the candidates are chosen: exact copies of train rows placed in different chunks
    (the nearest distance tied between chunks), and runner-ups in another chunk than the winner."""
    rng = np.random.default_rng(23)
    a = synth.sift_like(5000, rng)
    b = synth.sift_like(6000, rng)
    for k in range(300):
        b[13 * k] = a[11 * k]                    # d1 = 0
        if k % 3 == 0:
            b[5999 - 7 * k] = a[11 * k]          # the same row again, thousands of rows later
        else:
            t = a[11 * k].copy()
            t[k % 128] = t[k % 128] + 1 if t[k % 128] < 255 else 254
            b[5999 - 7 * k] = t                  # a runner-up one unit away, far from the winner
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        for ratio in ((4, 5), (1, 1), (1, 100)):
            for mutual in (True, False):
                g0, g1, gd = matcher.match_pair(s0, s1, ratio=ratio[0] / ratio[1], mutual=mutual, return_dist=True)
                o0, o1, od = orc.match_u8(a, b, orc.params(ratio, mutual))
                np.testing.assert_array_equal(g0, o0)
                np.testing.assert_array_equal(g1, o1)
                np.testing.assert_array_equal(gd, od)
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_mutual_check_spoiled_by_rows_that_are_no_candidates(matcher):
    """The verify pass looks only at image-0 rows whose own nearest distance does not exceed the largest
    accepted candidate distance of the pair.  Rows that spoil a candidate's mutual check without being
    candidates themselves (they fail the ratio test: two train rows equally near) must be among them, as must
    rows at exactly that distance, and ties in a candidate's column go to the lower image-0 row."""
    rng = np.random.default_rng(29)
    a = synth.sift_like(3000, rng)
    b = synth.sift_like(2600, rng)

    def bump(row, k, amount):
        out = row.copy()
        out[k % 128] = out[k % 128] + amount if out[k % 128] <= 255 - amount else out[k % 128] - amount
        return out

    for k in range(120):
        j, j2 = 20 * k, 20 * k + 7
        i, sp = 23 * k + 1, 23 * k + 5          # candidate row and its spoiler (a higher row: only distance decides)
        a[i] = bump(b[j], k, 3)                  # d(i, j) = 9: passes the ratio test
        if k % 3 == 0:
            b[j2] = bump(b[j], k + 1, 1)         # a second train row as near to the spoiler as the first ...
            a[sp] = bump(b[j], k + 2, 1)         # ... so the spoiler (d = 1 to j, 2 to j2) fails the ratio test itself
        elif k % 3 == 1:
            a[sp] = bump(b[j], k + 2, 3)         # the same distance as the candidate, higher row: does not spoil
        else:
            a[23 * k] = bump(b[j], k + 2, 3)     # the same distance, LOWER row: takes the column
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        for ratio in ((4, 5), (1, 2)):
            g0, g1, gd = matcher.match_pair(s0, s1, ratio=ratio[0] / ratio[1], mutual=True, return_dist=True)
            o0, o1, od = orc.match_u8(a, b, orc.params(ratio, True))
            np.testing.assert_array_equal(g0, o0)
            np.testing.assert_array_equal(g1, o1)
            np.testing.assert_array_equal(gd, od)
            n0, _, _ = orc.match_u8(a, b, orc.params(ratio, False))
            assert len(n0) > len(o0) + 30     # the mutual check did reject rows
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_short_row_list_of_the_mutual_check_on_full_range_bytes(matcher):
    """Few accepted candidates among full-range random bytes: the mutual check then streams a list of a few
    rows, and the positions past the end of the list must behave like padding rows whatever the sign of the
    (zero-point-shifted) products of real rows would be."""
    rng = np.random.default_rng(31)
    a = rng.integers(0, 256, (565, 128)).astype(np.uint8)
    b = rng.integers(0, 256, (5605, 128)).astype(np.uint8)
    b[rng.choice(5605, 40, replace=False)] = a[rng.choice(565, 40, replace=False)]
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        for cap in (-1, 8510):
            g0, g1, gd = matcher.match_pair(s0, s1, ratio=0.75, mutual=True, max_dist_sq=cap, return_dist=True)
            o0, o1, od = orc.match_u8(a, b, orc.params((3, 4), True, max_dist_sq=cap))
            np.testing.assert_array_equal(g0, o0)
            np.testing.assert_array_equal(g1, o1)
            np.testing.assert_array_equal(gd, od)
            assert len(o0) >= 30
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_parity_of_the_runner_up_decides(matcher):
    """d1 * den^2 < num^2 * d2 with d2 straddling the boundary: rows built so that the true runner-up
    distance is 2 k or 2 k + 1 around the value where the test flips."""
    a = np.zeros((64, 128), np.uint8)
    b = np.zeros((64 * 40, 128), np.uint8) + 200      # far filler
    want_pairs = 0
    for i in range(64):
        a[i, :] = 10
        a[i, i] = 60 + i                               # every query row distinct
        near = a[i].copy()
        near[(i + 1) % 128] += 2                       # d1 = 4
        b[40 * i + 3] = near
        run = a[i].copy()
        run[(i + 2) % 128] += 2                        # 4
        run[(i + 3) % 128] += 1 + (i & 1)              # + 1 or + 4: d2 = 5 or 8 ...
        run[(i + 5) % 128] += (i >> 1) & 1             # ... + 0 or 1
        b[40 * i + 3 + 32] = run                       # another tile
    for ratio in ((4, 5), (9, 10), (8, 9), (2, 3), (89, 100)):
        s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
        try:
            g0, g1, gd = matcher.match_pair(s0, s1, ratio=ratio[0] / ratio[1], mutual=False, return_dist=True)
            o0, o1, od = orc.match_u8(a, b, orc.params(ratio, False))
            np.testing.assert_array_equal(g0, o0)
            np.testing.assert_array_equal(g1, o1)
            np.testing.assert_array_equal(gd, od)
            want_pairs += len(o0)
        finally:
            matcher.release(s0)
            matcher.release(s1)
    assert want_pairs > 0


def test_distance_cap_with_ratio(matcher):
    a, b = synth.sift_pair(900, 1200, seed=19)
    s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
    try:
        for cap in (0, 1500, 20000, 10**9):
            g0, g1, gd = matcher.match_pair(s0, s1, ratio=0.8, mutual=True, max_dist_sq=cap, return_dist=True)
            o0, o1, od = orc.match_u8(a, b, orc.params((4, 5), True, max_dist_sq=cap))
            np.testing.assert_array_equal(g0, o0)
            np.testing.assert_array_equal(g1, o1)
            np.testing.assert_array_equal(gd, od)
    finally:
        matcher.release(s0)
        matcher.release(s1)


def test_pairs_of_many_rows_are_cut_into_slices(matcher):
    """Image-0 sets above 8192 rows: the select and row-list passes cut such a pair's rows into slices with a
    workgroup each, and the candidates must still come out in ascending row order.  Rows shared between the
    images put candidates into every slice; the list mixes the large pairs with small and empty ones, whose slices past
    the end of their rows do nothing."""
    rng = np.random.default_rng(4242)
    pool = synth.sift_like(40000, rng)
    sizes = [20000, 0, 9000, 1, 33000, 700]
    imgs = []
    for n in sizes:
        x = synth.sift_like(n, rng)
        k = n // 3
        if k:
            # every image holds noisy copies of its own choice of pool rows, each at most once
            x[rng.choice(n, k, replace=False)] = np.clip(
                pool[rng.choice(40000, k, replace=False)].astype(np.int32) + rng.integers(-3, 4, (k, 128)), 0, 255).astype(np.uint8)
        imgs.append(x)
    ids = [matcher.upload_u8(x) for x in imgs]
    pairs = [(0, 5), (1, 0), (4, 2), (3, 4), (2, 0), (5, 4), (4, 5), (0, 1)]
    try:
        for mutual in (True, False):
            off, i0, i1 = matcher.match_pairs([[ids[i], ids[j]] for i, j in pairs], ratio=0.8, mutual=mutual)
            for k, (i, j) in enumerate(pairs):
                o0, o1, _ = orc.match_u8(imgs[i], imgs[j], orc.params((4, 5), mutual))
                np.testing.assert_array_equal(i0[off[k]:off[k + 1]], o0)
                np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o1)
        g0, g1, gd = matcher.match_pair(ids[4], ids[0], ratio=0.8, mutual=True, return_dist=True)
        o0, o1, od = orc.match_u8(imgs[4], imgs[0], orc.params((4, 5), True))
        assert len(o0) > 1000
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
        np.testing.assert_array_equal(gd, od)
    finally:
        for s in ids:
            matcher.release(s)


def test_empty_sets_in_a_list_that_follows_a_full_one(matcher):
    """Eight pairs or more run the tail form of the query pass, whose strips leave per-strip candidate counts in
    scratch memory.  A pair with an empty train set has no strips at work: the counts a previous, full submission
    left at its place must not be read."""
    rng = np.random.default_rng(77)
    base = synth.sift_like(900, rng)
    imgs = []
    for n in (900, 700, 1100):
        x = synth.sift_like(n, rng)
        x[: n // 2] = np.clip(base[: n // 2].astype(np.int32) + rng.integers(-3, 4, (n // 2, 128)), 0, 255).astype(np.uint8)
        imgs.append(x)
    imgs.append(np.zeros((0, 128), np.uint8))
    ids = [matcher.upload_u8(x) for x in imgs]
    try:
        full = [(0, 1), (1, 2), (2, 0), (0, 2), (1, 0), (2, 1), (0, 1), (1, 2), (2, 0)]
        holes = [(0, 3), (3, 0), (1, 3), (0, 1), (3, 3), (2, 3), (3, 1), (1, 2), (2, 3)]
        for pairs in (full, holes, full, holes):
            off, i0, i1 = matcher.match_pairs([[ids[i], ids[j]] for i, j in pairs], ratio=0.8, mutual=True)
            for k, (i, j) in enumerate(pairs):
                o0, o1, _ = orc.match_u8(imgs[i], imgs[j], orc.params((4, 5), True))
                np.testing.assert_array_equal(i0[off[k]:off[k + 1]], o0)
                np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o1)
            assert off[-1] == len(i0)
    finally:
        for s in ids:
            matcher.release(s)


def _degenerate_images(rng):
    """Sets on which every nearest distance is tied: the all-ones dummy descriptors the reference's converter writes
    (FeaturePoints.cpp:358-367), and rows drawn from a pool of 40 with a third of them one unit off in one byte."""
    ones = [np.ones((n, 128), np.uint8) for n in (2048, 2048, 1500, 2048)]
    pool = synth.sift_like(40, rng)
    dups = []
    for n in (2048, 1800, 2048, 700):
        x = pool[rng.integers(0, 40, n)].copy()
        rows = rng.choice(n, n // 3, replace=False)
        x[rows, rng.integers(0, 128, len(rows))] ^= 1
        dups.append(np.ascontiguousarray(x))
    return ones, dups


def test_all_ones_and_duplicate_sets_take_the_rescan_pass(matcher, u8_mode):
    """Data full of exact duplicates: the tile-minimum form cannot settle a single row from its query pass, every row
    goes through the rescan pass (the packed form over gathered rows), and the lists still equal the oracle's."""
    rng = np.random.default_rng(101)
    for imgs in _degenerate_images(rng):
        ids = [matcher.upload_u8(x) for x in imgs]
        pairs = [(i, j) for i in range(4) for j in range(4) if i != j][:10]
        try:
            for mutual in (True, False):
                matcher.set_u8_mode(u8_mode)   # (forgets what the default mode learnt from the last list)
                off, i0, i1 = matcher.match_pairs([[ids[i], ids[j]] for i, j in pairs], ratio=0.8, mutual=mutual)
                rescanned = matcher.last_u8_rescanned()
                for k, (i, j) in enumerate(pairs):
                    o0, o1, _ = orc.match_u8(imgs[i], imgs[j], orc.params((4, 5), mutual))
                    np.testing.assert_array_equal(i0[off[k]:off[k + 1]], o0)
                    np.testing.assert_array_equal(i1[off[k]:off[k + 1]], o1)
                rows = sum(len(imgs[i]) for i, _ in pairs)
                if u8_mode == 1:
                    assert rescanned == 0 and matcher.last_u8_path() == 1
                else:
                    assert matcher.last_u8_path() == (2 if u8_mode == 3 else 0)   # (a short list of small sets is cut into chunks)
                    assert rescanned > rows // 2, "%d of %d rows rescanned" % (rescanned, rows)
            matcher.set_u8_mode(u8_mode)
            g0, g1, gd = matcher.match_pair(ids[0], ids[1], ratio=0.8, mutual=True, return_dist=True)
            o0, o1, od = orc.match_u8(imgs[0], imgs[1], orc.params((4, 5), True))
            np.testing.assert_array_equal(g0, o0)
            np.testing.assert_array_equal(g1, o1)
            np.testing.assert_array_equal(gd, od)
            if u8_mode == 3:   # (the default mode sends one small pair per call to the packed form)
                assert matcher.last_u8_rescanned() > len(imgs[0]) // 2
        finally:
            for s in ids:
                matcher.release(s)


def test_default_mode_goes_to_the_packed_form_after_a_submission_of_duplicates(matcher, u8_mode):
    """VSM_U8_AUTO: a submission in which more than one row in eight needed the rescan pass sends the next 31
    submissions through the packed form; the 32nd tries the tile-minimum form again.  Forced modes never change form."""
    rng = np.random.default_rng(102)
    ones, _ = _degenerate_images(rng)
    a, b = synth.sift_pair(600, 900, seed=103)
    ids = [matcher.upload_u8(x) for x in ones[:2]] + [matcher.upload_u8(a), matcher.upload_u8(b)]
    o0, o1, _ = orc.match_u8(a, b, orc.params((4, 5), True))
    tile = {0: 0, 1: 1, 3: 2}[u8_mode]
    try:
        matcher.match_pairs([[ids[0], ids[1]]] * 8, ratio=0.8, mutual=True)
        assert matcher.last_u8_path() == {0: 0, 1: 1, 3: 2}[u8_mode]
        assert matcher.last_u8_rescanned() == (0 if u8_mode == 1 else 8 * 2048)
        for k in range(33):
            off, g0, g1 = matcher.match_pairs([[ids[2], ids[3]]] * 2, ratio=0.8, mutual=True)
            np.testing.assert_array_equal(g0[:off[1]], o0)
            np.testing.assert_array_equal(g1[:off[1]], o1)
            want = 1 if (u8_mode == 0 and k < 31) else tile
            assert matcher.last_u8_path() == want, "submission %d after the duplicates ran form %d" % (k, matcher.last_u8_path())
    finally:
        for s in ids:
            matcher.release(s)
