"""Torch-free driver for hardware-counter passes (rocprofv3 --pmc) over the float pair-list kernels:
numpy-generated unit descriptors (a quarter of every image copied, with noise, from a shared pool),
uploaded through the C ABI, then a few submissions of `--pairs` pairs of the exhaustive i<j list."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visual_slam_amd as vsm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=24)
    ap.add_argument("--features", type=int, default=4096)
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--metric", choices=["dot", "l2"], default="dot")
    ap.add_argument("--ratio", type=float, default=0.0, help="Lowe ratio (L2 only); 0: none")
    ap.add_argument("--min-score", type=float, default=0.5, help="score threshold of the dot ranking")
    args = ap.parse_args()
    rng = np.random.default_rng(3)
    n = args.features
    pool = rng.standard_normal((n, 256), dtype=np.float32)
    m = vsm.Matcher(0)
    ids = []
    for _ in range(args.images):
        x = rng.standard_normal((n, 256), dtype=np.float32)
        k = n // 4
        x[rng.permutation(n)[:k]] = pool[rng.permutation(n)[:k]] + 0.05 * rng.standard_normal((k, 256), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        ids.append(m.upload_f32(x))
    allp = vsm.generate_exhaustive_pairs(args.images)[: args.pairs]
    pr = np.asarray([[ids[a], ids[b]] for a, b in allp], np.int32)
    m.set_timing(True)
    for _ in range(args.reps):
        if args.metric == "dot":
            tot = m.match_pairs_device(pr, ratio=None, mutual=True, metric=vsm.METRIC_DOT, min_score=args.min_score)
        else:
            tot = m.match_pairs_device(pr, ratio=(args.ratio or None), mutual=True, metric=vsm.METRIC_L2)
        m.fetch_matches(len(pr), tot)
    print("pairs %d matches %d kernels %.3f ms" % (len(pr), tot, m.last_kernel_ms()))
    m.close()


if __name__ == "__main__":
    main()
