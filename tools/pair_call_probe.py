"""Latency of one vsm_match_pair call (one SIFT pair of 8192 x 8192, lists to pageable host arrays) and of one float
pair call (4096 x 4096 x 256, mutual): the per-pair loop of the reference's exporter, call by call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visual_slam_amd as vsm  # noqa: E402
from visual_slam_amd import synth  # noqa: E402

m = vsm.Matcher(0)
a, b = synth.sift_pair(8192, 8192, seed=1)
s0, s1 = m.upload_u8(a), m.upload_u8(b)
for mode, name in ((vsm.U8_AUTO, "default"), (vsm.U8_TILE_SEPARATE, "tile-minimum"), (vsm.U8_PACKED, "packed")):
    m.set_u8_mode(mode)
    for _ in range(20):
        m.match_pair(s0, s1, ratio=0.8, mutual=True)
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        i0, i1 = m.match_pair(s0, s1, ratio=0.8, mutual=True)
    print("u8 8192 x 8192, %s: %.3f ms per call, %d matches" % (name, (time.perf_counter() - t0) / n * 1e3, len(i0)))
m.set_u8_mode(vsm.U8_AUTO)
fa, fb = synth.superpoint_pair(4096, 4096, 256, seed=2)
f0, f1 = m.upload_f32(fa), m.upload_f32(fb)
for _ in range(20):
    m.match_pair(f0, f1, ratio=None, mutual=True, metric=vsm.METRIC_DOT)
t0 = time.perf_counter()
for _ in range(n):
    i0, i1 = m.match_pair(f0, f1, ratio=None, mutual=True, metric=vsm.METRIC_DOT)
print("f32 4096 x 4096 x 256, mutual: %.3f ms per call, %d matches" % ((time.perf_counter() - t0) / n * 1e3, len(i0)))
m.close()
