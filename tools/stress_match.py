"""Longer randomised comparison of the GPU matcher with the CPU oracle than the test suite runs: data built to
hit ties and undecidable cases (descriptors quantised to a few values, many exact duplicates, clustered rows,
tiny value ranges), sizes across chunk and merge boundaries, both forms of the u8 query pass, every reverse-search
mode of the float pair path, single float pairs per call and raw float 2-NN searches.  Bit-exact or it stops.   usage: python tools/stress_match.py [--seeds N] [--first K]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as orc  # noqa: E402
import visual_slam_amd as vsm  # noqa: E402
from visual_slam_amd import synth  # noqa: E402


def u8_data(rng, n0, n1):
    kind = rng.integers(0, 7)
    if kind == 0:      # plain SIFT-like with planted matches
        a, b = synth.sift_pair(max(n0, 1), max(n1, 1), seed=int(rng.integers(1 << 30)), planted=float(rng.uniform(0, 0.7)),
                               noise=float(rng.uniform(1, 14)))
        return a[:n0], b[:n1]
    if kind == 1:      # few distinct values per component: ties everywhere
        q = int(rng.choice([2, 3, 5, 17]))
        a = (rng.integers(0, q, (n0, 128)) * (255 // (q - 1))).astype(np.uint8)
        b = (rng.integers(0, q, (n1, 128)) * (255 // (q - 1))).astype(np.uint8)
    elif kind == 2:    # clustered: rows are small perturbations of a few centres
        c = synth.sift_like(int(rng.integers(1, 12)), rng)
        a = np.clip(c[rng.integers(0, len(c), n0)].astype(np.int32) + rng.integers(-2, 3, (n0, 128)), 0, 255).astype(np.uint8)
        b = np.clip(c[rng.integers(0, len(c), n1)].astype(np.int32) + rng.integers(-2, 3, (n1, 128)), 0, 255).astype(np.uint8)
    elif kind == 3:    # heavy duplication across and inside the sets
        pool = synth.sift_like(max(int(rng.integers(1, 200)), 1), rng)
        a = pool[rng.integers(0, len(pool), n0)].copy()
        b = pool[rng.integers(0, len(pool), n1)].copy()
        for arr in (a, b):
            if len(arr):
                k = len(arr) // 3
                idx = rng.integers(0, len(arr), k)
                arr[idx, rng.integers(0, 128, k)] ^= 1        # one-unit differences: parity cases
    elif kind == 4:    # tiny value range / extremes
        lo = int(rng.choice([0, 120, 250]))
        a = rng.integers(lo, lo + 6, (n0, 128)).astype(np.uint8)
        b = rng.integers(lo, lo + 6, (n1, 128)).astype(np.uint8)
    elif kind == 5:    # full-range random bytes
        a = rng.integers(0, 256, (n0, 128)).astype(np.uint8)
        b = rng.integers(0, 256, (n1, 128)).astype(np.uint8)
    else:              # every row the same (the dummy descriptors of FeaturePoints.cpp:358-367), a few rows one unit off
        v = int(rng.choice([0, 1, 128, 255]))
        a = np.full((n0, 128), v, np.uint8)
        b = np.full((n1, 128), v, np.uint8)
        for arr in (a, b):
            k = len(arr) // int(rng.choice([3, 50, 10**9]))
            if k:
                arr[rng.integers(0, len(arr), k), rng.integers(0, 128, k)] ^= 1
    if n0 and n1 and rng.random() < 0.5:
        k = min(n0, n1, int(rng.integers(1, 60)))
        b[rng.integers(0, n1, k)] = a[rng.integers(0, n0, k)]
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def size(rng):
    r = rng.random()
    if r < 0.15:
        return int(rng.choice([0, 1, 2, 31, 32, 33, 255, 256, 257, 511, 512, 513]))
    if r < 0.85:
        return int(rng.integers(1, 3000))
    if r < 0.97:
        return int(rng.integers(3000, 9500))  # several chunks for a single pair, merged chunk records above 4096
    return int(rng.integers(9500, 40000))     # several slices of the select and row-list passes


def one_u8(m, rng):
    n0, n1 = size(rng), size(rng)
    a, b = u8_data(rng, n0, n1)
    num, den = ((int(rng.integers(1, 13)), 12) if rng.random() < 0.85 else (0, 1))
    mutual = bool(rng.random() < 0.7)
    cap = int(rng.integers(0, 150000)) if rng.random() < 0.25 else -1
    m.set_u8_mode(int(rng.choice([vsm.U8_AUTO, vsm.U8_TILE_FUSED, vsm.U8_TILE_FUSED, vsm.U8_TILE_SEPARATE, vsm.U8_PACKED])))
    s0, s1 = m.upload_u8(a), m.upload_u8(b)
    try:
        g0, g1, gd = m.match_pair(s0, s1, ratio=(num / den if num else None), mutual=mutual, max_dist_sq=cap, return_dist=True)
        o0, o1, od = orc.match_u8(a, b, orc.params((num, den) if num else None, mutual, max_dist_sq=cap))
        ok = np.array_equal(g0, o0) and np.array_equal(g1, o1) and np.array_equal(gd, od)
        return ok, "u8 n0=%d n1=%d ratio=%d/%d mutual=%d cap=%d matches=%d" % (n0, n1, num, den, mutual, cap, len(o0))
    finally:
        m.set_u8_mode(vsm.U8_AUTO)
        m.release(s0)
        m.release(s1)


def one_u8_list(m, rng):
    nimg = int(rng.integers(2, 7))
    imgs = []
    base_a, _ = u8_data(rng, 1500, 1)
    for _ in range(nimg):
        n = 0 if rng.random() < 0.08 else int(rng.integers(0, 2500))   # empty sets inside a list now and then
        d, _ = u8_data(rng, n, 1)
        k = min(n, len(base_a)) // 2
        if k:
            d[rng.choice(n, k, replace=False)] = base_a[rng.choice(len(base_a), k, replace=False)]
        imgs.append(np.ascontiguousarray(d))
    ids = [m.upload_u8(d) for d in imgs]
    pairs = [(int(rng.integers(0, nimg)), int(rng.integers(0, nimg))) for _ in range(int(rng.integers(1, 30)))]
    mutual = bool(rng.random() < 0.7)
    m.set_u8_mode(int(rng.choice([vsm.U8_AUTO, vsm.U8_TILE_FUSED, vsm.U8_TILE_SEPARATE, vsm.U8_PACKED])))
    try:
        off, i0, i1 = m.match_pairs([[ids[a], ids[b]] for a, b in pairs], ratio=0.8, mutual=mutual)
        ok = True
        for k, (a, b) in enumerate(pairs):
            o0, o1, _ = orc.match_u8(imgs[a], imgs[b], orc.params((4, 5), mutual))
            ok = ok and np.array_equal(i0[off[k]:off[k + 1]], o0) and np.array_equal(i1[off[k]:off[k + 1]], o1)
        return ok, "u8 list of %d pairs over %d images mutual=%d matches=%d" % (len(pairs), nimg, mutual, off[-1])
    finally:
        m.set_u8_mode(vsm.U8_AUTO)
        for s in ids:
            m.release(s)


def one_f32_list(m, rng):
    nimg = int(rng.integers(2, 6))
    dim = int(rng.choice([32, 64, 100, 128, 256]))
    pool = synth.unit_f32(800, dim, rng)
    imgs = []
    for _ in range(nimg):
        n = int(rng.integers(1, 1800))
        d = synth.unit_f32(n, dim, rng)
        k = min(n, 800) // 2
        if k:
            noisy = pool[rng.choice(800, k, replace=False)] + float(rng.uniform(0, 0.08)) * rng.standard_normal((k, dim)).astype(np.float32)
            d[rng.choice(n, k, replace=False)] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
        if rng.random() < 0.3 and n > 4:
            d[rng.integers(0, n, 4)] = d[0]                      # exact duplicates: ties
        imgs.append(np.ascontiguousarray(d, np.float32))
    ids = [m.upload_f32(d) for d in imgs]
    pairs = [(int(rng.integers(0, nimg)), int(rng.integers(0, nimg))) for _ in range(int(rng.integers(1, 20)))]
    metric = vsm.METRIC_DOT if rng.random() < 0.6 else vsm.METRIC_L2
    ratio = (3, 4) if metric == vsm.METRIC_L2 and rng.random() < 0.6 else None
    mutual = bool(rng.random() < 0.8)
    thr = float(rng.choice([-np.inf, 0.0, 0.3, 0.7, 0.95]))
    m.set_f32_reverse(int(rng.integers(0, 3)))
    try:
        off, i0, i1 = m.match_pairs(np.asarray([[ids[a], ids[b]] for a, b in pairs], np.int32), ratio=(0.75 if ratio else None),
                                    mutual=mutual, metric=metric, min_score=thr)
        ok = True
        for k, (a, b) in enumerate(pairs):
            o0, _, _, _ = orc.match_dense_f32(imgs[a], imgs[b], metric, orc.params(ratio, mutual, min_score=thr))
            sel = np.nonzero(o0 >= 0)[0]
            ok = ok and np.array_equal(i0[off[k]:off[k + 1]], sel) and np.array_equal(i1[off[k]:off[k + 1]], o0[sel])
        return ok, "f32 list of %d pairs dim=%d metric=%d ratio=%s mutual=%d thr=%g matches=%d" % (
            len(pairs), dim, metric, ratio, mutual, thr, off[-1])
    finally:
        m.set_f32_reverse(vsm.F32_REVERSE_AUTO)
        for s in ids:
            m.release(s)


def one_f32_pair(m, rng):
    """One float pair per call (its searches travel as kernel arguments, cut into up to 32 splits) and a raw 2-NN search
    over the same sets; a long run of such calls also exercises what one call leaves zeroed for the next."""
    dim = int(rng.choice([32, 64, 100, 128, 256]))
    r = rng.random()
    n0 = int(rng.integers(1, 600)) if r < 0.3 else int(rng.integers(600, 6000))
    n1 = int(rng.integers(1, 600)) if rng.random() < 0.3 else int(rng.integers(600, 7000))
    a, b = synth.unit_f32(n0, dim, rng), synth.unit_f32(n1, dim, rng)
    k = min(n0, n1) // 2
    if k:
        noisy = a[rng.choice(n0, k, replace=False)] + float(rng.uniform(0, 0.08)) * rng.standard_normal((k, dim)).astype(np.float32)
        b[rng.choice(n1, k, replace=False)] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
    if rng.random() < 0.3 and n1 > 40:
        b[rng.integers(0, n1, int(rng.integers(2, 40)))] = b[0]           # exact duplicates: ties, many survivors
    if rng.random() < 0.2 and n0 > 4:
        a[rng.integers(0, n0, 4)] = b[int(rng.integers(0, n1))]
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    metric = vsm.METRIC_DOT if rng.random() < 0.6 else vsm.METRIC_L2
    ratio = (3, 4) if metric == vsm.METRIC_L2 and rng.random() < 0.6 else None
    mutual = bool(rng.random() < 0.8)
    thr = float(rng.choice([-np.inf, -np.inf, 0.0, 0.3, 0.7]))
    m.set_f32_reverse(int(rng.integers(0, 3)))
    s0, s1 = m.upload_f32(a), m.upload_f32(b)
    try:
        prm = orc.params(ratio, mutual, min_score=thr)
        o0, o1, ob, _ = orc.match_dense_f32(a, b, metric, prm)
        sel = np.nonzero(o0 >= 0)[0]
        ok = True
        for _ in range(int(rng.integers(1, 4))):                           # the same call again: left-clean state
            g0, g1, gd = m.match_pair(s0, s1, ratio=(0.75 if ratio else None), mutual=mutual, metric=metric, min_score=thr,
                                      return_dist=True)
            ok = ok and np.array_equal(g0, sel) and np.array_equal(g1, o0[sel]) and np.array_equal(gd, ob[sel])
        d0, d1 = m.match_pair_dense(s0, s1, ratio=(0.75 if ratio else None), mutual=mutual, metric=metric, min_score=thr)
        ok = ok and np.array_equal(d0, o0) and np.array_equal(d1, o1)
        gi1, gi2, gk1, gk2 = m.knn2(s0, s1, metric, want_idx2=True)
        oi, ok1, ok2 = orc.knn2_f32(a, b, metric)
        ok = ok and np.array_equal(gi1, oi) and np.array_equal(gk1, ok1) and np.array_equal(gk2, ok2)
        return ok, "f32 pair n0=%d n1=%d dim=%d metric=%d ratio=%s mutual=%d thr=%g matches=%d" % (
            n0, n1, dim, metric, ratio, mutual, thr, len(sel))
    finally:
        m.set_f32_reverse(vsm.F32_REVERSE_AUTO)
        m.release(s0)
        m.release(s1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    m = vsm.Matcher(0)
    t0 = time.time()
    kinds = [one_u8, one_u8, one_u8_list, one_f32_list, one_f32_pair]
    for seed in range(a.first, a.first + a.seeds):
        rng = np.random.default_rng(70000 + seed)
        ok, what = kinds[seed % len(kinds)](m, rng)
        if not ok:
            print("MISMATCH seed %d: %s" % (seed, what), flush=True)
            return 1
        if seed % 20 == 0:
            print("seed %d ok (%.0f s): %s" % (seed, time.time() - t0, what), flush=True)
    print("all %d seeds equal to the oracle, %.0f s" % (a.seeds, time.time() - t0))
    m.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
