"""One SIFT pair of 65536 x 65536 rows, call by call: vsm_match_pair (one call, lists to pageable host arrays)
against vsm_match_pairs_device + vsm_fetch_matches (what a pair-list caller does)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visual_slam_amd as vsm  # noqa: E402
from visual_slam_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = vsm.Matcher(0)
a, b = synth.sift_pair(n, n, seed=1)
s0, s1 = m.upload_u8(a), m.upload_u8(b)
m.set_timing(True)
pr = np.asarray([[s0, s1]], np.int32)
reps = 200
for _ in range(20):
    m.match_pair(s0, s1, ratio=0.8, mutual=True)
t0 = time.perf_counter()
for _ in range(reps):
    i0, i1 = m.match_pair(s0, s1, ratio=0.8, mutual=True)
print("match_pair: %.3f ms per call, %d matches, kernel %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3, len(i0), m.last_kernel_ms()))
for _ in range(20):
    tot = m.match_pairs_device(pr, ratio=0.8, mutual=True)
    m.fetch_matches(1, tot)
t0 = time.perf_counter()
for _ in range(reps):
    tot = m.match_pairs_device(pr, ratio=0.8, mutual=True)
    off, j0, j1 = m.fetch_matches(1, tot)
print("match_pairs_device + fetch_matches: %.3f ms per call, %d matches, kernel %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3, tot, m.last_kernel_ms()))
assert np.array_equal(i0, j0) and np.array_equal(i1, j1)
m.close()
