"""Writes a directory of synthetic VisualSfM .sift files (<i>.sift, i = 0..N-1) for end-to-end runs of
vsm_match_pairs: SIFT-like u8 descriptors (|N(0,1)| components, L2-normalised to 512, rounded), 30 % of
every image's rows noisy copies of a shared pool so that pairs have real correspondences."""
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
    for i in range(a.images):
        x = raw(a.features)
        dst = rng.permutation(a.features)[:k]
        src = rng.permutation(a.features)[:k]
        x[dst] = pool[src] + 6.0 * rng.standard_normal((k, 128), dtype=np.float32)
        des = np.clip(np.rint(x), 0, 255).astype(np.uint8)
        loc = rng.random((a.features, 5), dtype=np.float32)
        host.write_sift(os.path.join(a.out, "%d.sift" % i), loc, des)
    print("wrote %d files of %d features to %s" % (a.images, a.features, a.out))


if __name__ == "__main__":
    main()
