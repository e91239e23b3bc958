"""Torch-free driver for hardware-counter passes (rocprofv3 --pmc) over the SIFT matching kernels.

The benchmark builds its inputs with torch kernels; under a counter-collecting profiler those
dispatches are instrumented too.  This probe keeps the process to the library's own kernels:
numpy-generated descriptors, uploaded through the C ABI, then a few submissions of the same shape
as one benchmark step (first `--pairs` pairs of the exhaustive i<j list).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visual_slam_amd as vsm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=300)
    ap.add_argument("--features", type=int, default=8192)
    ap.add_argument("--pairs", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    rng = np.random.default_rng(3)
    m = vsm.Matcher(0)
    ids = []
    for _ in range(args.images):
        # SIFT-like magnitudes: mostly small bins, a few large ones
        d = np.minimum(rng.gamma(0.8, 28.0, (args.features, 128)), 255.0).astype(np.uint8)
        ids.append(m.upload_u8(d))
    allp = vsm.generate_exhaustive_pairs(args.images)[: args.pairs]
    pr = np.asarray([[ids[a], ids[b]] for a, b in allp], np.int32)
    m.set_timing(True)
    times = []
    for _ in range(args.reps):
        tot = m.match_pairs_device(pr, ratio=0.8, mutual=True)
        m.fetch_matches(len(pr), tot)
        times.append(m.last_kernel_ms())
    times.sort()
    print("pairs %d matches %d query min %.3f median %.3f max %.3f ms (%d launches) verify %.3f ms"
          % (len(pr), tot, times[0], times[len(times) // 2], times[-1], len(times), m.last_verify_ms()))
    m.close()


if __name__ == "__main__":
    main()
