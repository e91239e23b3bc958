"""Torch-free driver for hardware-counter passes over the float database query: 16384 unit queries
against `--rows` unit rows of dimension 256 (numpy-generated), a few searches."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visual_slam_amd as vsm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--queries", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    rng = np.random.default_rng(3)
    q = rng.standard_normal((args.queries, 256), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    db = rng.standard_normal((args.rows, 256), dtype=np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    m = vsm.Matcher(0)
    sq, sd = m.upload_f32(q), m.upload_f32(db)
    m.set_timing(True)
    for _ in range(args.reps):
        m.knn2(sq, sd, vsm.METRIC_L2)
    print("kernels %.3f ms, path %d" % (m.last_kernel_ms(), m.last_f32_path()))
    m.close()


if __name__ == "__main__":
    main()
