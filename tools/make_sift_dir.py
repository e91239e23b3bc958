"""Writes a directory of synthetic VisualSfM .sift files (<i>.sift, i = 0..N-1) for end-to-end runs of
vsm_match_pairs: SIFT-like u8 descriptors (|N(0,1)| components, L2-normalised to 512, rounded), 30 % of
every image's rows noisy copies of a shared pool so that pairs have real correspondences; the pool rows
are points of one 3-D scene, so the correspondences obey a two-view geometry."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_slam_amd import host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--images", type=int, default=100)
    ap.add_argument("--features", type=int, default=8192)
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    rng = np.random.default_rng(a.seed)

    def raw(n):
        x = np.abs(rng.standard_normal((n, 128), dtype=np.float32))
        return x * (512.0 / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-6))

    pool = raw(a.features)
    k = int(0.3 * a.features)
    # every pool row is a point of one 3-D scene and every image a camera looking at it, so that the
    # matches of a pair obey a two-view geometry (for --verify runs); the other keypoints are random
    X = np.stack([rng.uniform(-4, 4, a.features), rng.uniform(-3, 3, a.features), rng.uniform(5, 12, a.features)], 1)
    w, h, f = 1600.0, 1200.0, 1200.0
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1.0]])
    for i in range(a.images):
        x = raw(a.features)
        dst = rng.permutation(a.features)[:k]
        src = rng.permutation(a.features)[:k]
        x[dst] = pool[src] + 6.0 * rng.standard_normal((k, 128), dtype=np.float32)
        des = np.clip(np.rint(x), 0, 255).astype(np.uint8)
        ang = rng.uniform(-0.2, 0.2, 3)
        ca, sa = np.cos(ang), np.sin(ang)
        R = (np.array([[ca[2], -sa[2], 0], [sa[2], ca[2], 0], [0, 0, 1.0]]) @ np.array([[ca[1], 0, sa[1]], [0, 1.0, 0], [-sa[1], 0, ca[1]]])
             @ np.array([[1.0, 0, 0], [0, ca[0], -sa[0]], [0, sa[0], ca[0]]]))
        t = rng.uniform(-1.0, 1.0, 3) * np.array([1.0, 1.0, 0.3])
        proj = (K @ (R @ X[src].T + t[:, None])).T
        loc = np.zeros((a.features, 5), np.float32)
        loc[:, 0] = rng.uniform(0, w, a.features)
        loc[:, 1] = rng.uniform(0, h, a.features)
        loc[dst, :2] = proj[:, :2] / proj[:, 2:3] + 0.4 * rng.standard_normal((k, 2))
        loc[:, 3] = rng.uniform(1.0, 8.0, a.features)
        host.write_sift(os.path.join(a.out, "%d.sift" % i), loc, des)
    print("wrote %d files of %d features to %s" % (a.images, a.features, a.out))


if __name__ == "__main__":
    main()
